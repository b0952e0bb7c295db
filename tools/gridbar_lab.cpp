// Lab: cost of a grid-wide barrier inside one persistent launch on MI355X (one workgroup per CU), with and without
// data hand-off between the phases.   hipcc --offload-arch=gfx950 -O3 -o tools/gridbar_lab tools/gridbar_lab.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr unsigned SPIN_LIMIT = 1u << 22;

// VAR 0: every thread fences its own stores; thread 0 arrives (release) and spins with acquire loads
// VAR 1: workgroup barrier, then thread 0 alone: release add, relaxed spin, one acquire fence at the end
// VAR 2: as 1, hierarchical: per-XCD counter (blockIdx & 7), the last arriver of the XCD adds to the global counter,
//        everyone spins on the global counter
// VAR 3: as 2, but the last arriver of the whole grid publishes a per-XCD generation word that the XCD's workgroups spin on
template <int VAR>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned& target, unsigned nwg, unsigned* err) {
    if (VAR == 0) __threadfence();
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        if (VAR == 0) {
            target += nwg;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
            }
        } else if (VAR == 1) {
            target += nwg;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            // ctr[0]: global count of XCD groups; ctr[32 * (1 + x)]: arrivals of XCD x; ctr[32 * (9 + x)]: generation of XCD x
            const unsigned x = blockIdx.x & 7, per = nwg >> 3;   // grid is a multiple of 8
            target += 1;
            const unsigned old = __hip_atomic_fetch_add(ctr + 32 * (1 + x), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == target * per) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);         // chain the XCD's releases into the global one
                const unsigned g = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (VAR == 3 && g + 1 == target * 8) {
#pragma unroll
                    for (int y = 0; y < 8; ++y)
                        __hip_atomic_store(ctr + 32 * (9 + y), target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const unsigned* flag = VAR == 3 ? ctr + 32 * (9 + x) : ctr;
            const unsigned want = VAR == 3 ? target : target * 8;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    __syncthreads();
    return ok;
}

// mode 0: barriers only.  mode 1: each phase every workgroup writes 4 KiB and then reads the 4 KiB its neighbour
// (blockIdx + 97) wrote in the previous phase, checking the contents.
template <int VAR>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* ctr, unsigned* err, unsigned* buf, int nbar, int mode,
                                                  unsigned* bad) {
    unsigned target = 0;
    const unsigned nwg = gridDim.x;
    unsigned nbad = 0;
    for (int p = 0; p < nbar; ++p) {
        if (mode == 1) {
            unsigned* mine = buf + ((size_t)(p & 1) * nwg + blockIdx.x) * 1024;
            for (int k = threadIdx.x; k < 1024; k += 256) mine[k] = p * 1000003u + blockIdx.x * 1024 + k;
        }
        if (!grid_barrier<VAR>(ctr, target, nwg, err)) break;
        if (mode == 1) {
            const unsigned nb = (blockIdx.x + 97) % nwg;
            const unsigned* theirs = buf + ((size_t)(p & 1) * nwg + nb) * 1024;
            for (int k = threadIdx.x; k < 1024; k += 256)
                if (theirs[k] != p * 1000003u + nb * 1024 + k) ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
    int nbar = argc > 1 ? atoi(argv[1]) : 100;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    unsigned *ctr, *err, *buf, *bad;
    CK(hipMalloc(&ctr, 4096));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&buf, (size_t)2 * ncu * 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int var = 0; var < 4; ++var)
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : {ncu / 2, ncu}) {
            float best = 1e9f;
            unsigned herr = 0, hbad = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(ctr, 0, 4096));
                CK(hipMemset(err, 0, 4));
                CK(hipMemset(bad, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (var == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(grid), dim3(256), 0, 0, ctr, err, buf, nbar, mode, bad);
                if (var == 1) hipLaunchKernelGGL(bar_kernel<1>, dim3(grid), dim3(256), 0, 0, ctr, err, buf, nbar, mode, bad);
                if (var == 2) hipLaunchKernelGGL(bar_kernel<2>, dim3(grid), dim3(256), 0, 0, ctr, err, buf, nbar, mode, bad);
                if (var == 3) hipLaunchKernelGGL(bar_kernel<3>, dim3(grid), dim3(256), 0, 0, ctr, err, buf, nbar, mode, bad);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
                if (herr) break;
            }
            printf("var %d mode %d grid %3d: %d barriers %.1f us -> %.2f us per barrier  (timeout flag %u, bad words %u)\n", var, mode,
                   grid, nbar, best * 1e3f, best * 1e3f / nbar, herr, hbad);
            if (herr) return 1;
        }
    return 0;
}
