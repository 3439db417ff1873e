import sys
p = 'tinyllama.cpp_amd/csrc/gten_decode_attn.h'
s = open(p).read()
old1 = "    const float ex = (c < n) ? expf(sc - mx) : 0.f;\n    const float sm = block_sum_n<4>(ex, red + 4);\n\n    // ---- probabilities against the chunk's own statistics, rounded to the activation dtype in registers\n    float pr = (c < n) ? ex / sm : 0.f;"
new1 = "    const float ex = (c < n) ? __expf(sc - mx) : 0.f;\n    const float sm = block_sum_n<4>(ex, red + 4);\n\n    // ---- probabilities against the chunk's own statistics, rounded to the activation dtype in registers\n    float pr = (c < n) ? ex * __builtin_amdgcn_rcpf(sm) : 0.f;"
if sys.argv[1] == 'apply':
    assert s.count(old1) == 1
    s = s.replace(old1, new1)
else:
    assert s.count(new1) == 1
    s = s.replace(new1, old1)
open(p, 'w').write(s)
