// tensor.h -- gten::Tensor with HBM-backed storage.
//
// Public interface follows the reference's gten/tensor.h:20-137 (same member
// names, shape/stride algebra, block-aware byte strides, shallow-copy
// semantics, resize-within-capacity, external-pointer constructor), so code
// written against the reference's Tensor compiles against this one.  What is
// different is underneath: storage lives in MI355X HBM (gten_hip_malloc instead
// of std::malloc, gten/tensor.cpp:61), allocated on first device use, with a
// host mirror that exists only for the three places host code touches tensor
// bytes in the reference: the checkpoint loader writing weights
// (tinyllama.cpp:320), the sampler reading logits (tinyllama.cpp:414,464) and
// token ids arriving as a host pointer (tinyllama.cpp:406,458).
//
//   data_ptr<T>()      host view; brings the mirror up to date (D2H) and, for
//                      the non-const overload, marks it as possibly modified
//   device_ptr()       HBM address for kernels; uploads a modified mirror first
//   device_weight()    same for Q8/Q4 weights, repacked once for the GPU
//                      (include/gten_hip.h, "Weight layouts")
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/gten_hip.h"
#include "gten_types.h"
#include "log.h"
#include "quants.h"

namespace gten {

// bytes requested by Tensor constructors (gten/tensor.h:17)
inline int64_t G_TensorMemAllocated = 0;

namespace detail {

// One process drives one GPU (replicas).  GTEN_HIP_DEVICE or, under
// torchrun-style launchers, LOCAL_RANK selects it.
inline void ensure_runtime()
{
    static bool ready = false;
    if (ready) return;
    int dev = 0;
    if (const char* e = std::getenv("GTEN_HIP_DEVICE")) dev = std::atoi(e);
    else if (const char* l = std::getenv("LOCAL_RANK")) dev = std::atoi(l);
    GTEN_HIP_OK(gten_hip_init(dev));
    ready = true;
}

// Deferred work (gten/modules.h: a single-row forward is recorded module by module and runs as ONE fused
// decoder step when lm_head is reached).  Anything that looks at tensor bytes -- a host view, a device pointer
// for another operator -- first lets the recorder materialise what it holds, operator by operator, so a caller
// that stops half way or inspects an intermediate tensor sees exactly what the reference would have computed.
inline void (*g_pending_settle)() = nullptr;
inline void settle_pending()
{
    if (g_pending_settle) g_pending_settle();
}

struct Storage {
    void* dev = nullptr;          // HBM, owned
    uint8_t* host = nullptr;      // host mirror (owned) or caller memory (external)
    size_t nbytes = 0;
    bool external = false;        // host points at caller memory we must not free
    bool host_newer = false;      // mirror holds bytes the device copy lacks
    bool dev_newer = false;       // device copy holds bytes the mirror lacks
    bool packed = false;          // device bytes are in the packed weight layout

    Storage() = default;
    Storage(const Storage&) = delete;
    Storage& operator=(const Storage&) = delete;
    ~Storage()
    {
        if (dev) gten_hip_free(dev);
        if (host && !external) std::free(host);
    }

    void need_dev()
    {
        if (dev) return;
        ensure_runtime();
        GTEN_HIP_OK(gten_hip_malloc(&dev, nbytes));
        GTEN_HIP_OK(gten_hip_memset(dev, 0, nbytes));
    }
    void need_host()
    {
        if (host) return;
        host = static_cast<uint8_t*>(std::calloc(nbytes ? nbytes : 1, 1));
        GTEN_ASSERTM(host, "Failed to allocate %zuMB of host memory.", nbytes / 1000000);
    }
    uint8_t* host_view(bool will_write)
    {
        settle_pending();
        GTEN_ASSERTM(!packed, "host access to a weight that has already been repacked into HBM is not supported");
        need_host();
        if (dev_newer) {
            GTEN_HIP_OK(gten_hip_memcpy_d2h(host, dev, nbytes));
            dev_newer = false;
        }
        if (will_write) host_newer = true;
        return host;
    }
    void* dev_view(bool will_write)
    {
        settle_pending();
        need_dev();
        if (host_newer) {
            GTEN_HIP_OK(gten_hip_memcpy_h2d(dev, host, nbytes));
            host_newer = false;
            // large uploads (f16 weights) do not keep a second copy in host RAM;
            // a later host access re-reads it from HBM
            if (!external && nbytes >= (1u << 20)) { std::free(host); host = nullptr; dev_newer = true; }
        }
        if (will_write) dev_newer = true;
        return dev;
    }
    // Q8/Q4 weights: block stream (host, as read from the .gten file) ->
    // packed planes in HBM, once; the host copy is dropped afterwards.
    void* dev_weight(int dtype_code_, int rows, int cols)
    {
        settle_pending();
        if (packed) return dev;
        need_dev();
        GTEN_ASSERTM(host && host_newer, "weight tensor was never filled from the host");
        void* staging = nullptr;
        GTEN_HIP_OK(gten_hip_malloc(&staging, nbytes));
        GTEN_HIP_OK(gten_hip_memcpy_h2d(staging, host, nbytes));
        GTEN_HIP_OK(gten_hip_pack_weight(staging, dtype_code_, rows, cols, dev));
        GTEN_HIP_OK(gten_hip_free(staging));   // synchronises the stream first
        host_newer = false;
        packed = true;
        if (!external) { std::free(host); host = nullptr; }
        return dev;
    }
};

} // namespace detail

class Tensor {
public:
    Tensor() = default;

    // Owning tensor.  Byte counts as gten/tensor.cpp:37-57: Q8 rounds the last
    // dimension up to whole 34-byte blocks, Q4 needs it to be a multiple of 32.
    Tensor(const std::vector<int>& shape, Dtype dtype) : dtype_{dtype}
    {
        validate_shape(shape);
        shape_ = shape;
        set_strides_from_shape(shape);
        numel_ = numel_from_shape(shape);
        size_t bytes;
        if (dtype == kQint8 && shape.size() != 1) {
            const int last = shape.back();
            const size_t rows = (size_t)numel_ / (size_t)last;
            bytes = rows * (size_t)((last + globs::q8_block_size - 1) / globs::q8_block_size) * sizeof(Q8Block);
        } else if (dtype == kQint4) {
            GTEN_ASSERT(ndims() == 2);
            GTEN_ASSERT(dimsize(1) % globs::q4_block_size == 0);
            bytes = (size_t)dimsize(0) * (size_t)(dimsize(1) / globs::q4_block_size) * sizeof(Q4Block);
        } else {
            bytes = (size_t)numel_ * (size_t)itemsize();
        }
        store_ = std::make_shared<detail::Storage>();
        store_->nbytes = bytes;
        storage_size_ = bytes;
        G_TensorMemAllocated += (int64_t)bytes;
    }

    // Non-owning view over caller HOST memory (gten/tensor.cpp:74-86); this is
    // how token ids reach the model (tinyllama.cpp:406).
    Tensor(const void* data_ptr, const std::vector<int>& shape, Dtype dtype) : dtype_{dtype}
    {
        GTEN_ASSERTM(data_ptr != nullptr, "Expected a non-null pointer but got a nullptr.");
        validate_shape(shape);
        shape_ = shape;
        set_strides_from_shape(shape);
        numel_ = numel_from_shape(shape);
        store_ = std::make_shared<detail::Storage>();
        store_->host = static_cast<uint8_t*>(const_cast<void*>(data_ptr));
        store_->external = true;
        store_->host_newer = true;
        store_->nbytes = dense_bytes();
        storage_size_ = 0;
    }

    Tensor(const Tensor&) = default;
    Tensor(Tensor&&) = default;
    Tensor& operator=(const Tensor&) = default;
    Tensor& operator=(Tensor&&) = default;

    // ---- host access (see header comment)
    template <typename T> T* data_ptr() { return reinterpret_cast<T*>(store_->host_view(true)); }
    template <typename T> const T* data_ptr() const { return reinterpret_cast<const T*>(store_->host_view(false)); }
    void* data_ptr() { return store_->host_view(true); }
    const void* data_ptr() const { return store_->host_view(false); }

    // ---- device access (not in the reference: this is the HBM side)
    const void* device_ptr() const { return store_->dev_view(false); }
    void* device_ptr_mut() { return store_->dev_view(true); }
    const void* device_weight() const
    {
        if (dtype_ == kQint8 || dtype_ == kQint4) {
            GTEN_ASSERT(is_2d());
            return store_->dev_weight(dtype_code(dtype_), shape_[0], shape_[1]);
        }
        return store_->dev_view(false);
    }
    bool is_host_external() const { return store_ && store_->external; }
    // identity of the storage this handle aliases (shallow copies share it, gten/tensor.h:24-29)
    const void* storage_id() const { return store_.get(); }
    const void* host_external_ptr() const { return store_->host; }

    Dtype dtype() const { return dtype_; }

    int itemsize() const
    {
        switch (dtype_) {
        case Dtype::Qint8: return 1;
        case Dtype::Int32: return 4;
        case Dtype::Float16: return 2;
        case Dtype::Float32: return 4;
        default: GTEN_ASSERT(false); return 4;
        }
    }

    bool is_quantized() const { return dtype_ == kQint8; }
    bool is_1d() const { return shape_.size() == 1; }
    bool is_2d() const { return shape_.size() == 2; }
    bool is_3d() const { return shape_.size() == 3; }
    int ndims() const { return (int)shape_.size(); }
    int numel() const { return numel_; }

    int dimsize(int i) const
    {
        GTEN_ASSERT(i < int(shape_.size()));
        return shape_[i];
    }
    int stride(int i) const
    {
        GTEN_ASSERT(i < int(strides_.size()));
        return strides_[i];
    }
    // Byte stride; quantized dtypes count whole blocks (gten/tensor.h:97-117).
    int bstride(int i) const
    {
        GTEN_ASSERT(i < int(strides_.size()));
        const int s = strides_[i];
        if (dtype_ == kQint4) return s == 1 ? 1 : (s / globs::q4_block_size) * (int)sizeof(Q4Block);
        if (dtype_ == kQint8) return s == 1 ? 1 : (s / globs::q8_block_size) * (int)sizeof(Q8Block);
        return s * itemsize();
    }

    size_t nbytes() const { return storage_size_; }
    const std::vector<int>& shape() const { return shape_; }
    bool shape_eq(const std::vector<int>& shape) const { return shape == shape_; }

    // Re-shape within the allocated capacity, no reallocation (gten/tensor.cpp:124-134).
    void resize(const std::vector<int>& new_shape)
    {
        validate_shape(new_shape);
        const size_t need = (size_t)numel_from_shape(new_shape) * (size_t)itemsize();
        GTEN_ASSERTM(need <= storage_size_, "The new shape provided %s with cap=%zu exceeds shape %s with cap=%zu.",
                     shape_to_str(new_shape).c_str(), need, shape_str().c_str(), storage_size_);
        shape_ = new_shape;
        set_strides_from_shape(new_shape);
        numel_ = numel_from_shape(new_shape);
    }

    void set_strides(const std::vector<int>& strides)
    {
        GTEN_ASSERTM(strides.size() == shape_.size(), "The given strides ndims must match shape ndims.");
        strides_ = strides;
    }

    // Shallow handle with a new contiguous shape (gten/tensor.cpp:158-169).
    Tensor view(const std::vector<int>& new_shape) const
    {
        validate_shape(new_shape);
        const int new_numel = numel_from_shape(new_shape);
        GTEN_ASSERTM(new_numel == numel_, "New shape numel `%d` must be equal with old shape numel `%d`.", new_numel, numel_);
        Tensor out = *this;
        out.shape_ = new_shape;
        out.set_strides_from_shape(new_shape);
        return out;
    }

    // In-place axis permutation; returns a shallow copy (gten/tensor.cpp:173-190).
    Tensor permute(const std::vector<int>& indices)
    {
        GTEN_ASSERTM(indices.size() == shape_.size(), "The dims of indices `%zu` given do not match the tensor dims `%zu`.",
                     indices.size(), shape_.size());
        std::vector<int> s(shape_.size()), t(shape_.size());
        for (size_t i = 0; i < indices.size(); i++) {
            s[i] = shape_[indices[i]];
            t[i] = strides_[indices[i]];
        }
        shape_ = s;
        strides_ = t;
        return *this;
    }

    std::string shape_str() const { return shape_to_str(shape_); }
    std::string strides_str() const { return shape_to_str(strides_); }

    void print_info() const
    {
        std::cout << "Tensor(shape=" << shape_str() << ", strides=" << strides_str() << ", dtype=" << dtype_str(dtype_)
                  << ", numel=" << numel_ << ", bytes=" << storage_size_ << ", where=HBM)\n";
    }

    // Debug print: first/last few values of each row, read back from HBM.
    void print() const
    {
        std::cout << *this << "\n";
    }

    // Raw storage bytes to a file (gten/tensor.cpp:232-237).
    void save(const std::string& path) const
    {
        std::ofstream f(path, std::ios::binary);
        GTEN_ASSERTM(f.is_open(), "Failed to open `%s` for writing.", path.c_str());
        f.write(reinterpret_cast<const char*>(data_ptr()), (std::streamsize)store_->nbytes);
    }

    friend std::ostream& operator<<(std::ostream& os, const Tensor& t)
    {
        os << "Tensor" << t.shape_str() << "[" << dtype_str(t.dtype_) << "](";
        const int show = t.numel_ < 8 ? t.numel_ : 8;
        for (int i = 0; i < show; i++) os << (i ? ", " : "") << t.flat_value(i);
        if (t.numel_ > show) os << ", ...";
        return os << ")";
    }

private:
    Dtype dtype_ = kFloat32;
    std::shared_ptr<detail::Storage> store_;
    size_t storage_size_ = 0;
    int numel_ = 0;
    std::vector<int> shape_;
    std::vector<int> strides_;

    static std::string shape_to_str(const std::vector<int>& v)
    {
        std::stringstream s;
        s << "(";
        for (size_t i = 0; i < v.size(); i++) s << v[i] << (i + 1 < v.size() ? ", " : "");
        s << ")";
        return s.str();
    }
    void validate_shape(const std::vector<int>& shape) const
    {
        GTEN_ASSERTM(shape.size() != 0, "The given shape is empty.");
        GTEN_ASSERTM(shape.size() <= 3, "Shape with dimensions > 3 not supported.");
        for (size_t i = 0; i < shape.size(); i++)
            GTEN_ASSERTM(shape[i] > 0, "The value of dimension %zu: %d of the given shape is invalid!", i, shape[i]);
    }
    void set_strides_from_shape(const std::vector<int>& shape)
    {
        strides_.assign(shape.size(), 1);
        for (int i = (int)shape.size() - 2; i >= 0; i--) strides_[i] = strides_[i + 1] * shape[i + 1];
    }
    static int numel_from_shape(const std::vector<int>& shape)
    {
        int n = 1;
        for (int s : shape) n *= s;
        return n;
    }
    size_t dense_bytes() const
    {
        if (dtype_ == kQint8 || dtype_ == kQint4) {
            const int last = shape_.back();
            return (size_t)(numel_ / last) * gten_hip_row_bytes(dtype_code(dtype_), last);
        }
        return (size_t)numel_ * (size_t)itemsize();
    }
    // value of the i-th element of a contiguous tensor (debug printing only)
    float flat_value(int i) const
    {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(data_ptr());
        switch (dtype_) {
        case Dtype::Int32: return (float)reinterpret_cast<const int32_t*>(p)[i];
        case Dtype::Float32: return reinterpret_cast<const float*>(p)[i];
        case Dtype::Float16: return fp16_to_fp32(reinterpret_cast<const Float16*>(p)[i]);
        case Dtype::Qint8: {
            const int last = shape_.back(), bpr = (last + 31) / 32;
            const int row = i / last, col = i % last;
            const Q8Block* b = reinterpret_cast<const Q8Block*>(p) + (size_t)row * bpr + col / 32;
            return (float)b->data[col % 32] * fp16_to_fp32(b->delta);
        }
        case Dtype::Qint4: {
            const int last = shape_.back(), bpr = last / 32;
            const int row = i / last, col = i % last;
            const Q4Block* b = reinterpret_cast<const Q4Block*>(p) + (size_t)row * bpr + col / 32;
            const int e = col % 32;
            const int nib = e < 16 ? (b->data[e] >> 4) : (b->data[e - 16] & 0x0f);
            return (float)(nib - 7) * fp16_to_fp32(b->delta);
        }
        }
        return 0.f;
    }
};

} // namespace gten
