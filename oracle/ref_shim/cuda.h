// TEST INFRASTRUCTURE ONLY (oracle/): a host-side stand-in for the CUDA execution model, so that the
// reference's own .cu kernels (read where they lie under /root/reference, never copied) can be compiled
// with g++ and run on the CPU to produce golden vectors.  Nothing in the product path includes this.
//
// Model: a kernel launch `k<<<grid, block>>>(args)` is rewritten on the fly (see build_ref.py) into
// `ref_launch(grid, block, [&]{ k(args); })`, which calls the kernel body once per (block, thread) with
// the CUDA built-in index variables set.  The reference kernels use no shared memory, no barriers and no
// warp intrinsics, so a serial sweep is an exact emulation; atomicAdd degenerates to `+=`
// in launch order.
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <c10/util/Half.h>

#define __global__
#define __device__
#define __host__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

template <class Body>
inline void ref_launch(dim3 grid, dim3 block, Body&& body) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned tz = 0; tz < block.z; ++tz)
        for (unsigned ty = 0; ty < block.y; ++ty)
        for (unsigned tx = 0; tx < block.x; ++tx) {
            threadIdx = dim3(tx, ty, tz);
            body();
        }
    }
}

// device math the kernels call unqualified
using std::min;
using std::max;
inline float __expf(float x) { return expf(x); }
inline float __sinf(float x) { return sinf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// half types: c10::Half carries IEEE binary16 storage with float arithmetic + round-to-nearest-even on
// assignment, which is what the device code's at::Half operators do as well.
typedef c10::Half __half;
struct __half2 { __half x, y; };

template <class T> inline T atomicAdd(T* addr, T v) { T old = *addr; *addr = old + v; return old; }
inline int atomicAdd(int* addr, unsigned v) { int old = *addr; *addr = old + (int)v; return old; }
inline __half2 atomicAdd(__half2* addr, __half2 v) {
    __half2 old = *addr;
    addr->x = (__half)((float)old.x + (float)v.x);
    addr->y = (__half)((float)old.y + (float)v.y);
    return old;
}
