"""The command line program (tinyllama.cpp_amd/host/tinyllama_cli.cpp): option handling of the reference's main()
(tinyllama.cpp:134-245) -- everything that needs no GPU."""
import os
import subprocess

import pytest

from __graft_entry__ import load_package


@pytest.fixture(scope="module")
def cli():
    pkg = load_package()
    pkg.build.build_all()
    path = pkg.build.HOST_CLI
    assert os.path.exists(path)
    return path


def run(cli, *args):
    return subprocess.run([cli, *args], capture_output=True, text=True, timeout=60)


def test_help(cli):
    r = run(cli, "--help")
    assert r.returncode == 0 and "USAGE" in r.stdout and "--npred" in r.stdout and "-q4" in r.stdout


def test_unknown_argument_and_bad_values(cli):
    r = run(cli, "--bogus")
    assert r.returncode != 0 and "Unknown argument" in r.stderr
    assert run(cli, "--npred", "0").returncode != 0                    # tinyllama.cpp:181-184
    assert run(cli, "--npred", "abc").returncode != 0
    assert run(cli, "--temp", "0").returncode != 0                     # tinyllama.cpp:199-202
    assert run(cli, "--topk", "40000").returncode != 0                 # tinyllama.cpp:217-220
    assert run(cli, "--npred").returncode != 0


def test_missing_files_are_reported(cli, tmp_path):
    r = run(cli, "-q4", "-p", "hi", "--model", str(tmp_path / "nope.gten"))
    assert r.returncode != 0 and "cannot open the checkpoint" in r.stderr
