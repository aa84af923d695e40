// Micro-benchmark: cost of a barrier among the workgroups of ONE XCD (32 CUs sharing an L2), the building block a
// persistent LSTM would need.  Every workgroup reads its XCC id, takes a ticket on that XCC,
// then runs ITERS rounds of { publish a value, arrive on the XCC's counter, spin until all of the XCC's workgroups
// arrived, read a peer's value }.  All spins are bounded (a stuck round sets an error flag and the kernel exits).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

struct Ctl {
    unsigned tickets[8];        // workgroups seen per XCC
    unsigned total;             // workgroups arrived in the set-up phase
    unsigned err;
    unsigned pad[6];
    unsigned counter[8][32];    // one arrival counter per XCC (own cache line)
};

template <int SCOPE>   // 0: agent-scope atomics / sc1 loads, 1: workgroup-scope atomics (executed in the XCD's L2) / sc0 loads
__global__ __launch_bounds__(256) void xcd_barrier_kernel(Ctl* ctl, float* vals, int iters, unsigned* xcc_of_wg) {
    __shared__ unsigned s_xcc, s_ticket, s_n;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7;      // HW_REG_XCC_ID[3:0]
        s_xcc = xcc;
        s_ticket = atomicAdd(&ctl->tickets[xcc], 1u);
        xcc_of_wg[blockIdx.x] = xcc;
        __threadfence();
        atomicAdd(&ctl->total, 1u);
        long spin = 0;
        while (atomicAdd(&ctl->total, 0u) < gridDim.x) {          // read-modify-write: performed at the memory side, never stale
            if (++spin > 4000000) { ctl->err = 1; break; }
        }
        __threadfence();
        s_n = atomicAdd(&ctl->tickets[xcc], 0u);
    }
    __syncthreads();
    const unsigned xcc = s_xcc, me = s_ticket, n = s_n;
    unsigned* cnt = &ctl->counter[xcc][0];
    float* myvals = vals + (size_t)xcc * 64 * 256;          // [64 slots][256 floats] per XCC
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // publish: every thread writes one float of this workgroup's slot
        myvals[(size_t)(me & 63) * 256 + threadIdx.x] = (float)(it + me);
        if (SCOPE == 0) __threadfence(); else __builtin_amdgcn_s_waitcnt(0);     // stores acknowledged by L2
        __syncthreads();
        if (threadIdx.x == 0) {
            if (SCOPE == 0) atomicAdd(cnt, 1u);
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned want = n * (unsigned)(it + 1);
            long spin = 0;
            while (true) {
                // scope 1: an agent-scope load misses the CU's L1 and is served by the XCD's L2, where the
                // workgroup-scope atomics of this XCD's workgroups were performed; scope 0: read-modify-write at the memory side
                const unsigned v = SCOPE == 0 ? atomicAdd(cnt, 0u) : __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= want) break;
                if (++spin > 2000000) { ctl->err = 2; break; }
            }
        }
        __syncthreads();
        if (ctl->err) return;
        // consume: read the value of the next workgroup of this XCC (must be this round's)
        const unsigned peer = (me + 1) % n;
        const float* pv = myvals + (size_t)(peer & 63) * 256 + threadIdx.x;
        const float got = __hip_atomic_load(pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got != (float)(it + peer)) atomicAdd(&ctl->pad[SCOPE], 1u);           // stale read
        acc += got;
    }
    if (acc == -1.f) vals[0] = acc;
}

int main() {
    Ctl* ctl; float* vals; unsigned* xcc_of_wg;
    CHECK(hipMalloc(&ctl, sizeof(Ctl)));
    CHECK(hipMalloc(&vals, 8 * 64 * 256 * sizeof(float)));
    CHECK(hipMalloc(&xcc_of_wg, 256 * sizeof(unsigned)));
    const int iters = 2000;
    for (int scope = 0; scope < 2; ++scope) {
        CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
        CHECK(hipMemset(vals, 0, 8 * 64 * 256 * sizeof(float)));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        if (scope == 0) hipLaunchKernelGGL(xcd_barrier_kernel<0>, dim3(256), dim3(256), 0, 0, ctl, vals, iters, xcc_of_wg);
        else hipLaunchKernelGGL(xcd_barrier_kernel<1>, dim3(256), dim3(256), 0, 0, ctl, vals, iters, xcc_of_wg);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        Ctl h;
        CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        std::vector<unsigned> x(256);
        CHECK(hipMemcpy(x.data(), xcc_of_wg, 256 * sizeof(unsigned), hipMemcpyDeviceToHost));
        int rr = 0;
        for (int i = 0; i < 256; ++i) rr += (x[i] == (unsigned)(i & 7));
        printf("scope %s: %.3f us per round, err %u, stale reads %u, tickets per XCC %u %u %u %u %u %u %u %u, wg i on xcc i%%8: %d/256\n",
               scope ? "workgroup(L2)" : "agent", 1e3f * ms / iters, h.err, h.pad[scope], h.tickets[0], h.tickets[1], h.tickets[2],
               h.tickets[3], h.tickets[4], h.tickets[5], h.tickets[6], h.tickets[7], rr);
    }
    return 0;
}
