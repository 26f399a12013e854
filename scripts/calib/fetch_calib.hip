// fetch_calib.hip — known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS PATTERN
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other patterns are
// uncalibrated).  Patterns = the ones this library's kernels use.  Build + run: scripts/calib/run_calib.sh (on the GPU box).
// Each kernel touches N_BYTES = 1.5 GiB (> the 256-MiB Infinity Cache); the program prints "<kernel> <true bytes>" per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void stream16_read(const float4* __restrict__ in, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void stream16_read_nt(const f4v* __restrict__ in, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const f4v v = __builtin_nontemporal_load(in + i); acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void stream4_read(const float* __restrict__ in, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 123.456f) sink[0] = acc;
}
// 48-byte records, consecutive lanes read consecutive records (3 x 16-byte loads at a 48-byte stride): the backward composite's
// survivor stream, k_emit / k_render_bwd record reads of consecutive pairs
__global__ void rec48_stream(const float4* __restrict__ in, size_t nrec, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) sink[0] = acc;
}
// 48-byte records at random positions (every record exactly once): the forward composite's list walk, k_gather_slots
__global__ void rec48_gather(const float4* __restrict__ in, const uint32_t* __restrict__ idx, size_t nrec, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx[i];
        const float4 a = in[3 * r], b = in[3 * r + 1], c = in[3 * r + 2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void stream16_write(float4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void stream16_write_nt(f4v* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(f4v{1.f, 2.f, 3.f, (float)i}, out + i);
}
__global__ void rec48_stream_write(float4* __restrict__ out, size_t nrec) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        out[3 * i] = make_float4(1.f, 2.f, 3.f, 4.f); out[3 * i + 1] = make_float4(5.f, 6.f, 7.f, 8.f); out[3 * i + 2] = make_float4(9.f, 1.f, 2.f, (float)i);
    }
}
// 48-byte records written at random positions + one validity byte each at a random position: the backward composite's partial records
__global__ void rec48_scatter_write(float4* __restrict__ out, const uint32_t* __restrict__ idx, size_t nrec) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx[i];
        out[3 * r] = make_float4(1.f, 2.f, 3.f, 4.f); out[3 * r + 1] = make_float4(5.f, 6.f, 7.f, 8.f); out[3 * r + 2] = make_float4(9.f, 1.f, 2.f, (float)i);
    }
}
__global__ void byte_scatter_write(uint8_t* __restrict__ out, const uint32_t* __restrict__ idx, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[(size_t)idx[i] * 4] = 1;
}

int main() {
    const size_t N_BYTES = (size_t)3 << 29;  // 1.5 GiB
    const size_t nrec = N_BYTES / 48;
    float4 *a, *b;
    float* sink;
    uint32_t* idx;
    CK(hipMalloc(&a, N_BYTES)); CK(hipMalloc(&b, N_BYTES)); CK(hipMalloc(&sink, 256)); CK(hipMalloc(&idx, nrec * 4));
    CK(hipMemset(a, 0, N_BYTES)); CK(hipMemset(b, 0, N_BYTES));
    {
        std::vector<uint32_t> h(nrec);
        std::iota(h.begin(), h.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(h.begin(), h.end(), rng);
        CK(hipMemcpy(idx, h.data(), nrec * 4, hipMemcpyHostToDevice));
    }
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 2; ++rep) {
        stream16_read<<<grid, block>>>(a, N_BYTES / 16, sink); printf("stream16_read %zu 0\n", N_BYTES);
        stream16_read_nt<<<grid, block>>>((const f4v*)a, N_BYTES / 16, sink); printf("stream16_read_nt %zu 0\n", N_BYTES);
        stream4_read<<<grid, block>>>((const float*)a, N_BYTES / 4, sink); printf("stream4_read %zu 0\n", N_BYTES);
        rec48_stream<<<grid, block>>>(a, nrec, sink); printf("rec48_stream %zu 0\n", nrec * 48);
        rec48_gather<<<grid, block>>>(a, idx, nrec, sink); printf("rec48_gather %zu 0\n", nrec * 48 + nrec * 4);
        stream16_write<<<grid, block>>>(b, N_BYTES / 16); printf("stream16_write 0 %zu\n", N_BYTES);
        stream16_write_nt<<<grid, block>>>((f4v*)b, N_BYTES / 16); printf("stream16_write_nt 0 %zu\n", N_BYTES);
        rec48_stream_write<<<grid, block>>>(b, nrec); printf("rec48_stream_write 0 %zu\n", nrec * 48);
        rec48_scatter_write<<<grid, block>>>(b, idx, nrec); printf("rec48_scatter_write %zu %zu\n", nrec * 4, nrec * 48);
        byte_scatter_write<<<grid, block>>>((uint8_t*)b, idx, nrec); printf("byte_scatter_write %zu %zu\n", nrec * 4, nrec);
        CK(hipDeviceSynchronize());
    }
    return 0;
}
