// Microbenchmark (round 4): how fast can a launch shaped like the fast kernel - one 768-thread block per CU, 12 waves -
// write 8 GB as 1 KB-per-instruction non-temporal stores, 16 KB per "unit" per wave, at scattered 16 KB chunks?
// Variants: stores per burst, filler work (dependent FMAs) between bursts, waves per block.
// build: hipcc -O3 --offload-arch=gfx950 tools/exp_store_stream.hip -o tools/exp_store_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double nt_pair __attribute__((ext_vector_type(2)));

// "balance" experiment: the same total filler per unit, spent before ONE burst of 16 stores, before each of four bursts of 4,
// or before each single store - does trickling the stores through the compute shorten the pass?  NOSTORE: compute alone.
template <int BURST, bool NOSTORE>
__global__ void balance_kernel(double* out, const int* perm, long n_units, int filler_per_unit) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (blockDim.x >> 6);
    lds[threadIdx.x] = 0.0;
    double x = 1.0 + lane, y = 2.0 + lane, z = 3.0 + lane;      // three chains: three waves per SIMD keep the FMA pipe busy
    const int per = filler_per_unit * BURST / 16;
    for (long u = wave; u < n_units; u += stride) {
        double* base = out + (long)perm[u] * 2048 + 2 * lane;
#pragma unroll
        for (int b0 = 0; b0 < 16; b0 += BURST) {
            for (int f = 0; f < per; ++f) { x = fma(x, 1.0000001, 1e-9); y = fma(y, 0.9999999, 1e-9); z = fma(z, 1.0000002, 1e-9); }
            nt_pair v; v.x = x + y; v.y = z;
            if (!NOSTORE) {
#pragma unroll
                for (int b = 0; b < BURST; ++b)
                    __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(base + (b0 + b) * 128));
            }
        }
    }
    if (x + y + z == 123.456) out[0] = x;
}

template <int BURST>
__global__ void stream_kernel(double* out, const int* perm, long n_units, int filler, int lds_touch) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (blockDim.x >> 6);
    if (lds_touch) lds[threadIdx.x] = 0.0;
    double x = 1.0 + lane;
    for (long u = wave; u < n_units; u += stride) {
        double* base = out + (long)perm[u] * 2048 + 2 * lane;          // 16 KB chunk
#pragma unroll
        for (int b0 = 0; b0 < 16; b0 += BURST) {
            for (int f = 0; f < filler; ++f) x = fma(x, 1.0000001, 1e-9);        // dependent chain: ~8 cycles each
            nt_pair v; v.x = x; v.y = x;
#pragma unroll
            for (int b = 0; b < BURST; ++b)
                __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(base + (b0 + b) * 128));
        }
    }
    if (x == 123.456) out[0] = x;
}

// the stream kernel's shape (persistent, 16 KB units at random chunks) with other store instructions: 8 bytes per lane
// (32 instructions per unit) and raw buffer stores
template <int MODE>
__global__ void stream_alt_kernel(double* out, const int* perm, long n_units, long n_bytes) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (blockDim.x >> 6);
    lds[threadIdx.x] = 0.0;
    const double x = 1.0 + lane;
    // buffer resource over the first 2 GB of the allocation (raw, stride 0)
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
    for (long u = wave; u < n_units; u += stride) {
        const long chunk = perm[u];
        if (MODE == 0) {                    // 8 bytes per lane: 32 stores of 512 B
            double* base = out + chunk * 2048 + lane;
#pragma unroll
            for (int b = 0; b < 32; ++b) __builtin_nontemporal_store(x, base + b * 64);
        } else {                            // buffer_store_dwordx4 with a 32-bit offset (chunks within the first 2 GB only)
            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
            v4u v; v.x = v.y = v.z = v.w = static_cast<unsigned int>(lane);
            const int off = static_cast<int>((chunk % 131072) * 16384 + lane * 16);
#pragma unroll
            for (int b = 0; b < 16; ++b) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off + b * 1024, 0, 2 /* slc: streaming */);
        }
    }
}

// memset-like: a non-persistent grid, each thread writes VEC consecutive 16-byte pieces; PLAIN: ordinary stores
template <int VEC, bool PLAIN>
__global__ void fill_kernel(double* out, long n_pairs) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    nt_pair v; v.x = 1.0; v.y = 2.0;
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        if (i + k < n_pairs) {
            if (PLAIN) reinterpret_cast<nt_pair*>(out)[i + k] = v;
            else __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(out) + i + k);
        }
}
// persistent, but the WAVES of the whole grid advance together through the buffer: wave w writes 1 KB at (step * n_waves + w) KB
template <bool PLAIN>
__global__ void march_kernel(double* out, long n_kb) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (blockDim.x >> 6);
    nt_pair v; v.x = 1.0; v.y = 2.0;
    for (long kb = wave; kb < n_kb; kb += stride) {
        nt_pair* at = reinterpret_cast<nt_pair*>(out + kb * 128) + lane;
        if (PLAIN) *at = v; else __builtin_nontemporal_store(v, at);
    }
}

// persistent, XCD-aware: the blocks of XCD x = blockIdx % 8 write only the pieces (of `piece_kb` KB) whose index is
// congruent to x + rot mod 8 - the address pattern a round-robin grid of one-piece blocks produces by itself
__global__ void xcd_kernel(double* out, long n_pieces, int piece_kb, int rot) {
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    const long wx = (long)(blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);       // wave within the XCD
    const long wstride = (long)(gridDim.x >> 3) * (blockDim.x >> 6);
    nt_pair v; v.x = 1.0; v.y = 2.0;
    const int res = (xcd + rot) & 7;
    for (long m = wx; 8 * m + res < n_pieces; m += wstride) {
        double* base = out + (8 * m + res) * (long)piece_kb * 128 + 2 * lane;
        for (int k = 0; k < piece_kb; ++k) __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(base + k * 128));
    }
}

// persistent with a global ticket: a wave takes the next chunk of `chunk_kb` KB from an atomic counter just before it writes
__global__ void ticket_kernel(double* out, long n_chunks, int chunk_kb, unsigned long long* counter) {
    const int lane = threadIdx.x & 63;
    nt_pair v; v.x = 1.0; v.y = 2.0;
    while (true) {
        unsigned long long c = 0;
        if (lane == 0) c = atomicAdd(counter, 1ull);
        c = __shfl(c, 0, 64);
        if ((long)c >= n_chunks) break;
        double* base = out + (long)c * chunk_kb * 128 + 2 * lane;
        for (int k = 0; k < chunk_kb; ++k) __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(base + k * 128));
    }
}
// not persistent: block b writes the contiguous range of `per_block_kb` KB at b * per_block_kb, its waves striding through it
__global__ void range_kernel(double* out, long per_block_kb) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    lds[threadIdx.x] = 0.0;
    nt_pair v; v.x = 1.0; v.y = 2.0;
    double* base = out + (long)blockIdx.x * per_block_kb * 128 + 2 * lane;
    for (long kb = wave; kb < per_block_kb; kb += nw) __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(base + kb * 128));
}

int main(int argc, char** argv) {
    const long n_units = 500000;                       // x 16 KB = 8.19 GB
    double* out; int* perm;
    hipMalloc(&out, n_units * 16384);
    hipMalloc(&perm, n_units * sizeof(int));
    std::vector<int> h(n_units);
    for (long i = 0; i < n_units; ++i) h[i] = (int)i;
    srand(1);
    for (long i = n_units - 1; i > 0; --i) { long j = ((long)rand() * RAND_MAX + rand()) % (i + 1); std::swap(h[i], h[j]); }
    std::vector<int> ident(n_units);
    for (long i = 0; i < n_units; ++i) ident[i] = (int)i;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int n_cu = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); n_cu = prop.multiProcessorCount;
    auto run = [&](const char* label, int burst, int threads, int blocks_per_cu, int filler, size_t lds, bool random) {
        hipMemcpy(perm, random ? h.data() : ident.data(), n_units * sizeof(int), hipMemcpyHostToDevice);
        void (*k)(double*, const int*, long, int, int) = burst == 16 ? stream_kernel<16> : burst == 8 ? stream_kernel<8> : burst == 4 ? stream_kernel<4> : stream_kernel<1>;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(n_cu * blocks_per_cu), dim3(threads), lds, 0, out, perm, n_units, filler, 1);
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(n_cu * blocks_per_cu), dim3(threads), lds, 0, out, perm, n_units, filler, 1);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-64s %7.3f ms  %6.2f TB/s\n", label, ms, n_units * 16384.0 / ms / 1e9);
    };
    const size_t big = 159 * 1024;
    run("12 waves/CU (LDS-limited), burst 16, random chunks", 16, 768, 1, 0, big, true);
    run("12 waves/CU, burst 16, sequential chunks", 16, 768, 1, 0, big, false);
    run("12 waves/CU, burst 4, random", 4, 768, 1, 0, big, true);
    run("12 waves/CU, burst 1, random", 1, 768, 1, 0, big, true);
    run("12 waves/CU, burst 4, filler 40 per burst, random", 4, 768, 1, 40, big, true);
    run("12 waves/CU, burst 4, filler 200 per burst, random", 4, 768, 1, 200, big, true);
    run("12 waves/CU, burst 16, filler 800 per unit, random", 16, 768, 1, 800, big, true);
    run("16 waves/CU (1024 threads), burst 16, random", 16, 1024, 1, 0, big, true);
    run("8 waves/CU (512 threads), burst 16, random", 16, 512, 1, 0, big, true);
    run("4 waves/CU (256 threads), burst 16, random", 16, 256, 1, 0, big, true);
    run("32 waves/CU (2 x 1024, no LDS), burst 16, random", 16, 1024, 2, 0, 1024, true);
    run("32 waves/CU (2 x 1024, no LDS), burst 16, sequential", 16, 1024, 2, 0, 1024, false);
    auto timeit = [&](const char* label, auto&& launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-64s %7.3f ms  %6.2f TB/s\n", label, ms, n_units * 16384.0 / ms / 1e9);
    };
    const long n_pairs = n_units * 1024, n_kb = n_units * 16;
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(out, 0, n_units * 16384, 0); });
    timeit("fill: grid of 256-thread blocks, 1 x 16 B per thread, nt", [&] { hipLaunchKernelGGL((fill_kernel<1, false>), dim3((n_pairs + 255) / 256), dim3(256), 0, 0, out, n_pairs); });
    timeit("fill: 256-thread blocks, 2 x 16 B per thread, nt", [&] { hipLaunchKernelGGL((fill_kernel<2, false>), dim3((n_pairs / 2 + 255) / 256), dim3(256), 0, 0, out, n_pairs); });
    timeit("fill: 256-thread blocks, 1 x 16 B per thread, plain", [&] { hipLaunchKernelGGL((fill_kernel<1, true>), dim3((n_pairs + 255) / 256), dim3(256), 0, 0, out, n_pairs); });
    timeit("fill: 256-thread blocks, 2 x 16 B per thread, plain", [&] { hipLaunchKernelGGL((fill_kernel<2, true>), dim3((n_pairs / 2 + 255) / 256), dim3(256), 0, 0, out, n_pairs); });
    timeit("march: 256 x 768 persistent, waves advance together, nt", [&] { hipLaunchKernelGGL((march_kernel<false>), dim3(n_cu), dim3(768), 0, 0, out, n_kb); });
    timeit("march: 256 x 768 persistent, waves advance together, plain", [&] { hipLaunchKernelGGL((march_kernel<true>), dim3(n_cu), dim3(768), 0, 0, out, n_kb); });
    timeit("march: 2048 x 256 persistent, plain", [&] { hipLaunchKernelGGL((march_kernel<true>), dim3(n_cu * 8), dim3(256), 0, 0, out, n_kb); });
    {
        hipMemcpy(perm, h.data(), n_units * sizeof(int), hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void*)stream_alt_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)stream_alt_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        timeit("12 waves/CU, random chunks, 8 B per lane (32 x 512 B stores per unit), nt", [&] { hipLaunchKernelGGL(stream_alt_kernel<0>, dim3(n_cu), dim3(768), big, 0, out, perm, n_units, n_units * 16384); });
        timeit("12 waves/CU, random chunks (first 2 GB), buffer_store_dwordx4 slc", [&] { hipLaunchKernelGGL(stream_alt_kernel<1>, dim3(n_cu), dim3(768), big, 0, out, perm, n_units, n_units * 16384); });
    }
    {   // balance experiment
        hipMemcpy(perm, h.data(), n_units * sizeof(int), hipMemcpyHostToDevice);
        auto bal = [&](const char* label, auto kernel, int filler) {
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            timeit(label, [&] { hipLaunchKernelGGL(kernel, dim3(n_cu), dim3(768), big, 0, out, perm, n_units, filler); });
        };
        for (int filler : {160, 320, 480}) {
            char l0[96], l1[96], l2[96], l3[96];
            snprintf(l0, sizeof l0, "balance: filler %d per unit, compute alone", filler);
            snprintf(l1, sizeof l1, "balance: filler %d per unit, ONE burst of 16 stores", filler);
            snprintf(l2, sizeof l2, "balance: filler %d per unit, four bursts of 4", filler);
            snprintf(l3, sizeof l3, "balance: filler %d per unit, sixteen single stores", filler);
            bal(l0, balance_kernel<16, true>, filler);
            bal(l1, balance_kernel<16, false>, filler);
            bal(l2, balance_kernel<4, false>, filler);
            bal(l3, balance_kernel<1, false>, filler);
        }
    }
    unsigned long long* counter; hipMalloc(&counter, 8);
    for (int chunk_kb : {16})
        for (int threads : {768, 256}) {
            char label[96];
            snprintf(label, sizeof label, "ticket: persistent %d-thread blocks, chunks of %d KB", threads, chunk_kb);
            const long n_chunks = n_units * 16 / chunk_kb;
            const int blocks = threads == 768 ? n_cu : n_cu * 8;
            timeit(label, [&] { hipMemsetAsync(counter, 0, 8, 0); hipLaunchKernelGGL(ticket_kernel, dim3(blocks), dim3(threads), 0, 0, out, n_chunks, chunk_kb, counter); });
        }
    hipFuncSetAttribute((const void*)range_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (long per_block_kb : {12L, 96L, 768L, 6144L}) {
        char label[96];
        snprintf(label, sizeof label, "ranges: 768-thread blocks (159 KB LDS), %ld KB per block, %ld blocks", per_block_kb, n_kb / per_block_kb);
        timeit(label, [&] { hipLaunchKernelGGL(range_kernel, dim3(n_kb / per_block_kb), dim3(768), big, 0, out, per_block_kb); });
    }
    for (int piece_kb : {8})
        for (int rot : {0}) {
            char label[96];
            snprintf(label, sizeof label, "xcd-aware persistent 256 x 768: pieces of %d KB, rotation %d", piece_kb, rot);
            const long n_pieces = n_units * 16 / piece_kb;
            timeit(label, [&] { hipLaunchKernelGGL(xcd_kernel, dim3(n_cu), dim3(768), 0, 0, out, n_pieces, piece_kb, rot); });
        }
    return 0;
}
