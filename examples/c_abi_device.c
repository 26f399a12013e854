/* Device-level C caller of libs360.so — what a pybind / C++ binding of upstream's rasterize_gaussians /
 * rasterize_gaussians_backward (the functions behind /root/reference/src/model/decoder/cuda_splatting.py:113-124) would do:
 * hipMalloc the buffers, s360_layout -> s360_forward -> s360_backward on a stream, copy the results back.  No Python, no
 * torch: plain C99 against include/s360.h and the HIP runtime's C API.
 *
 *   gcc -std=c99 -Iinclude -I/opt/rocm/include examples/c_abi_device.c -Lsplatter360_amd -ls360 -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/splatter360_amd -Wl,-rpath,/opt/rocm/lib -o c_abi_device
 *   ./c_abi_device scene.bin out.bin
 *
 * scene.bin (written by tests/test_gpu_c_caller.py): int32 header {magic 0x53333630, P, V, H, W, M, sh_degree, flags,
 * max_instances}, then float32 arrays views[V*44], means[P*3], cov[P*(flags & COV9 ? 9 : 6)], opacities[P], shs[P*M*3],
 * dL_dimages[V*3*H*W].  out.bin: images[V*3*H*W], radii[V*P] (int32), d_means[P*3], d_cov, d_opacities[P], d_shs[P*M*3],
 * uint32 {num_instances, overflow}.  The test compares out.bin byte for byte with the same call made through the Python binding. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "s360.h"

#define HIP_OK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
            return 10;                                                                    \
        }                                                                                 \
    } while (0)
#define S360_CALL(x)                                                                      \
    do {                                                                                  \
        int rc_ = (x);                                                                    \
        if (rc_ != S360_OK) {                                                             \
            fprintf(stderr, "%s: %s\n", #x, s360_error_string(rc_));                      \
            return 11;                                                                    \
        }                                                                                 \
    } while (0)

static float* read_floats(FILE* f, size_t n) {
    float* p = (float*)malloc(n ? n * sizeof(float) : 4);
    if (!p || fread(p, sizeof(float), n, f) != n) {
        fprintf(stderr, "scene file truncated\n");
        exit(12);
    }
    return p;
}

static int upload(void** d, const void* h, size_t bytes) {
    HIP_OK(hipMalloc(d, bytes ? bytes : 4));
    if (bytes) HIP_OK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
}

static int download(FILE* f, const void* d, size_t bytes) {
    void* h = malloc(bytes ? bytes : 4);
    if (!h) return 13;
    if (bytes) HIP_OK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    fwrite(h, 1, bytes, f);
    free(h);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]);
        return 2;
    }
    if (s360_abi_version() != S360_ABI_VERSION) {
        fprintf(stderr, "ABI mismatch: header %d, library %d\n", S360_ABI_VERSION, s360_abi_version());
        return 3;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 4;
    int32_t hdr[9];
    if (fread(hdr, 4, 9, f) != 9 || hdr[0] != 0x53333630) return 5;
    S360Params prm;
    memset(&prm, 0, sizeof prm);
    prm.P = hdr[1]; prm.V = hdr[2]; prm.H = hdr[3]; prm.W = hdr[4]; prm.M = hdr[5]; prm.sh_degree = hdr[6];
    prm.flags = (uint32_t)hdr[7]; prm.max_instances = (uint32_t)hdr[8];
    const size_t P = (size_t)prm.P, V = (size_t)prm.V, npix = (size_t)prm.H * prm.W, ncov = (prm.flags & S360_FLAG_COV9) ? 9 : 6;
    float* h_views = read_floats(f, V * 44);
    float* h_means = read_floats(f, P * 3);
    float* h_cov = read_floats(f, P * ncov);
    float* h_op = read_floats(f, P);
    float* h_sh = read_floats(f, P * prm.M * 3);
    float* h_dimg = read_floats(f, V * 3 * npix);
    fclose(f);
    if (sizeof(S360View) != 44 * sizeof(float)) return 6;

    S360Layout lay;
    S360_CALL(s360_layout(&prm, &lay));
    void *d_views, *d_means, *d_cov, *d_op, *d_sh, *d_dimg, *images, *radii, *ws, *bws, *g_means, *g_cov, *g_op, *g_sh;
    if (upload(&d_views, h_views, V * 44 * 4) || upload(&d_means, h_means, P * 12) || upload(&d_cov, h_cov, P * ncov * 4) ||
        upload(&d_op, h_op, P * 4) || upload(&d_sh, h_sh, P * prm.M * 12) || upload(&d_dimg, h_dimg, V * 3 * npix * 4))
        return 7;
    HIP_OK(hipMalloc(&images, V * 3 * npix * 4));
    HIP_OK(hipMalloc(&radii, V * P * 4 + 4));
    HIP_OK(hipMalloc(&ws, lay.total_bytes));
    HIP_OK(hipMalloc(&bws, lay.backward_bytes));
    HIP_OK(hipMalloc(&g_means, P * 12 + 4));
    HIP_OK(hipMalloc(&g_cov, P * ncov * 4 + 4));
    HIP_OK(hipMalloc(&g_op, P * 4 + 4));
    HIP_OK(hipMalloc(&g_sh, P * prm.M * 12 + 4));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    S360_CALL(s360_forward(&prm, (const S360View*)d_views, (const float*)d_means, (const float*)d_cov, (const float*)d_op, (const float*)d_sh,
                           NULL, (float*)images, (int32_t*)radii, ws, lay.total_bytes, (void*)st));
    S360_CALL(s360_backward(&prm, (const S360View*)d_views, (const float*)d_means, (const float*)d_cov, (const float*)d_op,
                            (const float*)d_sh, NULL, ws, lay.total_bytes, (const float*)d_dimg, NULL, NULL, 0, (float*)g_means, NULL,
                            (float*)g_cov, (float*)g_op, (float*)g_sh, NULL, bws, lay.backward_bytes, (void*)st));
    HIP_OK(hipStreamSynchronize(st));

    FILE* o = fopen(argv[2], "wb");
    if (!o) return 8;
    if (download(o, images, V * 3 * npix * 4) || download(o, radii, V * P * 4) || download(o, g_means, P * 12) ||
        download(o, g_cov, P * ncov * 4) || download(o, g_op, P * 4) || download(o, g_sh, P * prm.M * 12) ||
        download(o, (const char*)ws + lay.header, 8))
        return 9;
    fclose(o);
    uint32_t head[2];
    HIP_OK(hipMemcpy(head, (const char*)ws + lay.header, 8, hipMemcpyDeviceToHost));
    printf("c caller: P %d V %d %dx%d num_instances %u overflow %u forward_ws %zu backward_ws %zu\n", prm.P, prm.V, prm.W, prm.H, head[0],
           head[1], lay.total_bytes, lay.backward_bytes);
    return head[1] ? 20 : 0;
}
