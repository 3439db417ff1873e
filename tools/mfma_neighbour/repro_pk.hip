// translation unit 1 of the reproducer: plain -O3 (packed f32 allowed)
#define VICTIM_PACKED 1
#define VICTIM k_victim_pk
#include "repro_victim.inc"
