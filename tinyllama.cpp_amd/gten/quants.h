// quants.h -- block formats of the gten API (gten/quants.h:12-31 of the
// reference).  On the MI355X path these structs describe BYTES IN HBM
// (activations, KV cache) and in .gten files; all arithmetic on them happens in
// the HIP kernels behind include/gten_hip.h, so there are no host quantizers
// here.
#pragma once

#include "gten_types.h"

namespace gten {

namespace globs {
static const int q8_block_size = 32;
static const int q4_block_size = 32;
}

#pragma pack(push, 1)
struct Q8Block {
    Float16 delta;
    Qint8 data[globs::q8_block_size];
};
struct Q4Block {
    Float16 delta;
    Qint4 data[globs::q4_block_size / 2];
};
#pragma pack(pop)

static_assert(sizeof(Q8Block) == 34, "Q8 block is 34 bytes");
static_assert(sizeof(Q4Block) == 18, "Q4 block is 18 bytes");

} // namespace gten
