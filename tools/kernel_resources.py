#!/usr/bin/env python3
"""Register / scratch / LDS use of every gfx950 kernel in an object or shared library built by hipcc
(tools/kernel_resources.py tinyllama.cpp_amd/csrc/libgten_hip.so [name filter]): unbundles the fat binary's code
objects and reads their kernel descriptors' metadata.  Used to check that shipped kernels do not spill."""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def code_objects(path):
    data = open(path, "rb").read()
    out, pos = [], 0
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        n = int.from_bytes(data[i + 24:i + 32], "little")
        q = i + 32
        for _ in range(n):
            off, size, tlen = (int.from_bytes(data[q + 8 * k:q + 8 * k + 8], "little") for k in range(3))
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + 24


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([LLVM + "llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, e) or [None, "0"])[1]
            rows.append((g("name"), int(g("vgpr_count")), int(g("sgpr_count")), int(g("vgpr_spill_count")),
                         int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("vgpr sgpr spill scratchB ldsB  kernel")
    for r, nm in sorted(zip(rows, names), key=lambda x: x[1]):
        if flt in nm:
            print("%4d %4d %5d %8d %5d  %s" % (r[1], r[2], r[3], r[4], r[5], nm[:170]))


if __name__ == "__main__":
    main()
