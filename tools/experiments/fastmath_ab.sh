cd $GRAFT_REPO_ROOT
run() { for i in 1 2 3; do python3 bench.py --brief --steps 64 --warmup 16 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['family_avg_launch_us']['decode_attn_score'])"; done; }
run base
python3 tools/experiments/fastmath_patch.py apply && python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/fm_build.log 2>&1 || { tail gpurun_out/fm_build.log; exit 1; }
run fastmath
python3 tools/experiments/fastmath_patch.py revert
