/* A plain-C host of the multi-device pool (include/rover_fe.h, rfe_pool_*): what a C / C++ program such as Rover-SLAM
 * (src/Tracking.cc:645-651 constructs its extractors in C++) would write to run configs[3] without Python.
 * usage: pool_driver frames.u8 F H W Kmax members transport sp.rfew lg.rfew out.bin
 *   frames.u8: F frames of H x W bytes; members: pool size, all on device 0 (one-GPU box); transport: 0 auto / 1 rccl / 2 copy
 *   out.bin: n [F] i32 | S [F-1] i32 | kxy [F,K,2] i32 | pairs [F-1,K,2] i32 | ms [F-1,K] f32
 * Compiled as C99 by the CPU test (the header is C-clean), run on the GPU box by tests/test_pool.py. */
#include <stdio.h>
#include <stdlib.h>
#include "rover_fe.h"

int main(int argc, char** argv) {
    if (argc != 11) { fprintf(stderr, "usage: pool_driver frames.u8 F H W Kmax members transport sp.rfew lg.rfew out.bin\n"); return 2; }
    const int F = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), K = atoi(argv[5]), members = atoi(argv[6]), transport = atoi(argv[7]);
    const size_t fb = (size_t)F * H * W;
    uint8_t* frames = (uint8_t*)malloc(fb);
    FILE* fi = fopen(argv[1], "rb");
    if (!fi || fread(frames, 1, fb, fi) != fb) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(fi);
    int devs[64];
    for (int i = 0; i < members && i < 64; ++i) devs[i] = 0;
    rfe_pool* pool = NULL;
    if (rfe_pool_create(devs, members, &pool) != RFE_OK) { fprintf(stderr, "rfe_pool_create: %s\n", rfe_pool_last_error(NULL)); return 1; }
    if (rfe_pool_load_weights(pool, argv[8], argv[9]) != RFE_OK) { fprintf(stderr, "weights: %s\n", rfe_pool_last_error(pool)); return 1; }
    const int P = F - 1;
    int32_t* n = (int32_t*)calloc(F, 4);
    int32_t* S = (int32_t*)calloc(P > 0 ? P : 1, 4);
    int32_t* kxy = (int32_t*)calloc((size_t)F * K * 2, 4);
    int32_t* pairs = (int32_t*)calloc((size_t)(P > 0 ? P : 1) * K * 2, 4);
    float* ms = (float*)calloc((size_t)(P > 0 ? P : 1) * K, 4);
    const int rc = rfe_pool_extract_match_stream(pool, frames, H, W, W, F, K, 0.0005f, 0.1f, transport, n, kxy, NULL, NULL, S, pairs, ms);
    if (rc != RFE_OK) { fprintf(stderr, "stream (%d): %s\n", rc, rfe_pool_last_error(pool)); return 1; }
    for (int r = 0; r < rfe_pool_size(pool); ++r) {
        int first, fr, own;
        rfe_pool_shard(F, rfe_pool_size(pool), r, &first, &fr, &own);
        printf("member %d: frames [%d, %d), %d pairs\n", r, first, first + fr, own);
    }
    FILE* fo = fopen(argv[10], "wb");
    if (!fo) return 2;
    fwrite(n, 4, F, fo); fwrite(S, 4, P, fo); fwrite(kxy, 4, (size_t)F * K * 2, fo); fwrite(pairs, 4, (size_t)P * K * 2, fo); fwrite(ms, 4, (size_t)P * K, fo);
    fclose(fo);
    rfe_pool_destroy(pool);
    free(frames); free(n); free(S); free(kxy); free(pairs); free(ms);
    return 0;
}
