// cu_mask_probe.hip -- does a CU-masked stream keep a kernel off the masked CUs on this box, and how are the mask bits laid out?
//   hipcc --offload-arch=gfx950 -O2 -o tools/cu_mask_probe tools/cu_mask_probe.hip && tools/cu_mask_probe
// Each workgroup records (XCC_ID, SE, CU) from the hardware-id registers.  Pass 1: no mask.  Pass 2: a stream created with
// hipExtStreamCreateWithCUMask and the first `keep` bits set.  Pass 3: a whole-CU-LDS kernel on an unmasked stream while a
// long LDS-filling kernel runs on the masked stream -- do the big workgroups start at once (the masked-off CUs are free)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include <map>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(unsigned* out, int spin) {
    __shared__ double pad[4096];                            // 32 KB: up to 5 per CU
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID, low 4 bits
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    pad[threadIdx.x] = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(4);
    if (pad[threadIdx.x] < 0) out[0] = 0;
}

__global__ void k_fill(long long* stamps, int spin) {      // 40 KB of LDS: 4 per CU fill it
    __shared__ double pad[5120];
    pad[threadIdx.x] = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(4);
    if (pad[threadIdx.x] < 0) stamps[1] = 0;
    if (threadIdx.x == 0) atomicMax((unsigned long long*)&stamps[1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

__global__ void k_big(long long* stamps, unsigned* where) {   // 150 KB of LDS: needs an empty CU
    __shared__ double pad[19200];
    pad[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0) {
        stamps[2 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    }
    if (pad[threadIdx.x] < 0) stamps[0] = 0;
}

static void summarize(const char* tag, const std::vector<unsigned>& v, int n) {
    std::map<unsigned, int> per_cu;
    std::set<unsigned> xccs;
    for (int i = 0; i < n; ++i) {
        unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
        xccs.insert(xcc);
    }
    printf("%s: %d workgroups on %zu distinct CUs, %zu XCCs\n", tag, n, per_cu.size(), xccs.size());
    std::map<unsigned, std::set<unsigned>> by_xcc;
    for (auto& kv : per_cu) by_xcc[kv.first >> 16].insert(kv.first & 0xffff);
    for (auto& kv : by_xcc) {
        printf("  xcc %u: %zu CUs:", kv.first, kv.second.size());
        for (unsigned c : kv.second) printf(" %u.%u.%u", (c >> 8) & 0xf, (c >> 4) & 1, c & 0xf);
        printf("\n");
    }
}

int main(int argc, char** argv) {
    int keep = argc > 1 ? atoi(argv[1]) : 192;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const int nwg = 2048;
    unsigned* dw;
    CHK(hipMalloc(&dw, sizeof(unsigned) * 2 * nwg));
    std::vector<unsigned> hw(2 * nwg);
    hipStream_t s0, sm;
    CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, s0, dw, 2000);
    CHK(hipStreamSynchronize(s0));
    CHK(hipMemcpy(hw.data(), dw, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost));
    summarize("unmasked", hw, nwg);

    uint32_t mask[8];
    memset(mask, 0, sizeof mask);
    for (int i = 0; i < keep; ++i) mask[i / 32] |= 1u << (i % 32);
    hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask);
    printf("hipExtStreamCreateWithCUMask(first %d bits): %s\n", keep, hipGetErrorString(e));
    if (e != hipSuccess) return 2;
    hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, sm, dw, 2000);
    CHK(hipStreamSynchronize(sm));
    CHK(hipMemcpy(hw.data(), dw, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost));
    summarize("masked", hw, nwg);

    // pass 3: fill the masked stream's CUs for ~300 us, then launch 40 whole-CU workgroups on the unmasked stream
    long long* ds;
    CHK(hipMalloc(&ds, sizeof(long long) * 64));
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipMemset(ds, 0, sizeof(long long) * 64));
        hipLaunchKernelGGL(k_fill, dim3(4 * keep), dim3(256), 0, sm, ds, 30000);     // 300 us at 100 MHz
        hipLaunchKernelGGL(k_big, dim3(40), dim3(256), 0, s0, ds, dw);
        CHK(hipDeviceSynchronize());
        long long st[64];
        CHK(hipMemcpy(st, ds, sizeof st, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(hw.data(), dw, sizeof(unsigned) * 2 * 40, hipMemcpyDeviceToHost));
        long long lo = st[2], hi = st[2];
        for (int i = 0; i < 40; ++i) { if (st[2 + i] < lo) lo = st[2 + i]; if (st[2 + i] > hi) hi = st[2 + i]; }
        printf("rep %d: fill kernel %.1f .. %.1f us; big workgroups started %.1f .. %.1f us after the fill kernel's start\n", rep, 0.0,
               (st[1] - st[0]) / 100.0, (lo - st[0]) / 100.0, (hi - st[0]) / 100.0);
        if (rep == 1) summarize("big workgroups", hw, 40);
    }
    // pass 4: the same without the mask (fill on an ordinary stream, 4 x keep workgroups): where do the big ones go?
    hipStream_t s1;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHK(hipMemset(ds, 0, sizeof(long long) * 64));
    hipLaunchKernelGGL(k_fill, dim3(4 * keep), dim3(256), 0, s1, ds, 30000);
    hipLaunchKernelGGL(k_big, dim3(40), dim3(256), 0, s0, ds, dw);
    CHK(hipDeviceSynchronize());
    long long st[64];
    CHK(hipMemcpy(st, ds, sizeof st, hipMemcpyDeviceToHost));
    long long lo = st[2], hi = st[2];
    for (int i = 0; i < 40; ++i) { if (st[2 + i] < lo) lo = st[2 + i]; if (st[2 + i] > hi) hi = st[2 + i]; }
    printf("no mask: fill kernel lasted %.1f us; big workgroups started %.1f .. %.1f us after its start\n", (st[1] - st[0]) / 100.0,
           (lo - st[0]) / 100.0, (hi - st[0]) / 100.0);
    return 0;
}
