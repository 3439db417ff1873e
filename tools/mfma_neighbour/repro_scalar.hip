// translation unit 2 of the reproducer: -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops
#define VICTIM_PACKED 0
#define VICTIM k_victim_scalar
#include "repro_victim.inc"
