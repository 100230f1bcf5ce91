/*
 * rfe_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See rfe_oracle.h.
 *
 * PARITY UNPINNED against the true reference (onnxruntime + the two missing .onnx blobs);
 * cross-checked against `transformers` SuperPoint/LightGlue modelling code (tests/golden/).
 *
 * Build: gcc -O2 -ftree-vectorize -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC
 * (-ffp-contract=off: every fused multiply-add in the canonical arithmetic is an explicit fmaf)
 */
#include "rfe_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define MINI(a, b) ((a) < (b) ? (a) : (b))
#define MAXI(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------ */
/* SuperPoint layer table: published architecture, names as in the     */
/* reference's dead libtorch header include/SuperPoint.h:24-41         */
/* ------------------------------------------------------------------ */
typedef struct { int cin, cout, k; } sp_layer_t;
static const sp_layer_t SP_LAYERS[12] = {
    {1, 64, 3},    /* conv1a */ {64, 64, 3},   /* conv1b (+pool) */
    {64, 64, 3},   /* conv2a */ {64, 64, 3},   /* conv2b (+pool) */
    {64, 128, 3},  /* conv3a */ {128, 128, 3}, /* conv3b (+pool) */
    {128, 128, 3}, /* conv4a */ {128, 128, 3}, /* conv4b */
    {128, 256, 3}, /* convPa */ {256, 65, 1},  /* convPb */
    {128, 256, 3}, /* convDa */ {256, 256, 1}, /* convDb */
};

int64_t rfo_sp_layer_offset(int layer, int want_bias) {
    int64_t off = 0;
    for (int l = 0; l < 12; ++l) {
        int64_t wn = (int64_t)SP_LAYERS[l].cout * SP_LAYERS[l].cin * SP_LAYERS[l].k * SP_LAYERS[l].k;
        if (l == layer) return want_bias ? off + wn : off;
        off += wn + SP_LAYERS[l].cout;
    }
    return off;
}
int64_t rfo_sp_weight_count(void) { return rfo_sp_layer_offset(12, 0); }

/* ------------------------------------------------------------------ */
/* canonical expf: Cody-Waite reduction + degree-6 polynomial, fmaf only */
/* ------------------------------------------------------------------ */
float rfo_expf(float x) {
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    float n = nearbyintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return y * s.f;
}

/* ------------------------------------------------------------------ */
/* GEMM-like micro kernel: out[m][n] = bias[n] then fmaf chain over k  */
/* rows of A are addressed through a callback-free (base,stride) list  */
/* ------------------------------------------------------------------ */
#define NB 16
#define MB 4

/* transposed+padded weights: wt[k][Npad], Npad multiple of NB */
static float* transpose_pad(const float* w, int N, int K, int* npad_out) {
    int Npad = (N + NB - 1) / NB * NB;
    float* wt = (float*)calloc((size_t)K * Npad, sizeof(float));
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) wt[(size_t)k * Npad + n] = w[(size_t)n * K + k];
    *npad_out = Npad;
    return wt;
}

void rfo_linear(const float* a, int M, int K, const float* w, const float* bias, int N, float* out) {
    int Npad;
    float* wt = transpose_pad(w, N, K, &Npad);
#pragma omp parallel for schedule(static)
    for (int m0 = 0; m0 < M; m0 += MB) {
        int mb = MINI(MB, M - m0);
        for (int n0 = 0; n0 < Npad; n0 += NB) {
            float acc[MB][NB];
            for (int p = 0; p < MB; ++p)
                for (int j = 0; j < NB; ++j) acc[p][j] = (bias && n0 + j < N) ? bias[n0 + j] : 0.0f;
            for (int k = 0; k < K; ++k) {
                const float* wr = wt + (size_t)k * Npad + n0;
                for (int p = 0; p < mb; ++p) {
                    float av = a[(size_t)(m0 + p) * K + k];
#pragma omp simd
                    for (int j = 0; j < NB; ++j) acc[p][j] = fmaf(av, wr[j], acc[p][j]);
                }
            }
            int nb = MINI(NB, N - n0);
            for (int p = 0; p < mb; ++p)
                for (int j = 0; j < nb; ++j) out[(size_t)(m0 + p) * N + n0 + j] = acc[p][j];
        }
    }
    free(wt);
}

/* NHWC conv3x3, pad 1.  kappa = ci*9 + ky*3 + kx (PyTorch OIHW row order). */
void rfo_conv3x3(const float* in, int H, int W, int Cin, const float* w, const float* bias,
                 int Cout, int relu, int pool, float* out) {
    const int K = Cin * 9;
    int Npad;
    float* wt = transpose_pad(w, Cout, K, &Npad);
    const int Hp = H + 2, Wp = W + 2;
    float* inp = (float*)calloc((size_t)Hp * Wp * Cin, sizeof(float));
    for (int y = 0; y < H; ++y)
        memcpy(inp + ((size_t)(y + 1) * Wp + 1) * Cin, in + (size_t)y * W * Cin, sizeof(float) * W * Cin);
    float* full = pool ? (float*)malloc(sizeof(float) * (size_t)H * W * Cout) : out;
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < H; ++y) {
        for (int x0 = 0; x0 < W; x0 += MB) {
            int mb = MINI(MB, W - x0);
            for (int n0 = 0; n0 < Npad; n0 += NB) {
                float acc[MB][NB];
                for (int p = 0; p < MB; ++p)
                    for (int j = 0; j < NB; ++j) acc[p][j] = (n0 + j < Cout) ? bias[n0 + j] : 0.0f;
                for (int ci = 0; ci < Cin; ++ci)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const float* wr = wt + (size_t)(ci * 9 + ky * 3 + kx) * Npad + n0;
                            for (int p = 0; p < mb; ++p) {
                                float av = inp[((size_t)(y + ky) * Wp + (x0 + p + kx)) * Cin + ci];
#pragma omp simd
                                for (int j = 0; j < NB; ++j) acc[p][j] = fmaf(av, wr[j], acc[p][j]);
                            }
                        }
                int nb = MINI(NB, Cout - n0);
                for (int p = 0; p < mb; ++p)
                    for (int j = 0; j < nb; ++j) {
                        float v = acc[p][j];
                        if (relu) v = v > 0.0f ? v : 0.0f;
                        full[((size_t)y * W + x0 + p) * Cout + n0 + j] = v;
                    }
            }
        }
    }
    if (pool) {
        int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x)
                for (int c = 0; c < Cout; ++c) {
                    const float* b = full + ((size_t)(2 * y) * W + 2 * x) * Cout + c;
                    float v0 = fmaxf(b[0], b[Cout]);
                    float v1 = fmaxf(b[(size_t)W * Cout], b[(size_t)W * Cout + Cout]);
                    out[((size_t)y * Wo + x) * Cout + c] = fmaxf(v0, v1);
                }
        free(full);
    }
    free(inp);
    free(wt);
}

void rfo_softmax65_d2s(const float* logits, int Hc, int Wc, float* score) {
    const int W = Wc * 8;
#pragma omp parallel for schedule(static)
    for (int cell = 0; cell < Hc * Wc; ++cell) {
        const float* l = logits + (size_t)cell * 65;
        float m = l[0];
        for (int c = 1; c < 65; ++c) m = fmaxf(m, l[c]);
        float e[65], s = 0.0f;
        for (int c = 0; c < 65; ++c) { e[c] = rfo_expf(l[c] - m); s = s + e[c]; }
        int cy = cell / Wc, cx = cell % Wc;
        for (int c = 0; c < 64; ++c)
            score[(size_t)(cy * 8 + (c >> 3)) * W + cx * 8 + (c & 7)] = e[c] / s;
    }
}

/* separable (2r+1)^2 max pool, stride 1, implicit -inf padding */
static void maxpool_sq(const float* in, int H, int W, int r, float* tmp, float* out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float m = -INFINITY;
            for (int d = MAXI(0, x - r); d <= MINI(W - 1, x + r); ++d) m = fmaxf(m, in[(size_t)y * W + d]);
            tmp[(size_t)y * W + x] = m;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float m = -INFINITY;
            for (int d = MAXI(0, y - r); d <= MINI(H - 1, y + r); ++d) m = fmaxf(m, tmp[(size_t)d * W + x]);
            out[(size_t)y * W + x] = m;
        }
}

/* simple_nms exactly as the published SuperPoint/LightGlue post-processing:
 *   max_mask = s == maxpool(s)
 *   2x { supp = maxpool(max_mask) > 0; s' = supp ? 0 : s; new = s' == maxpool(s');
 *        max_mask |= new & ~supp }
 *   out = max_mask ? s : 0 */
void rfo_nms(const float* score, int H, int W, int radius, float* out) {
    size_t n = (size_t)H * W;
    float* tmp = (float*)malloc(sizeof(float) * n);
    float* mp = (float*)malloc(sizeof(float) * n);
    float* mask = (float*)malloc(sizeof(float) * n);
    float* supp = (float*)malloc(sizeof(float) * n);
    float* ss = (float*)malloc(sizeof(float) * n);
    maxpool_sq(score, H, W, radius, tmp, mp);
    for (size_t i = 0; i < n; ++i) mask[i] = (score[i] == mp[i]) ? 1.0f : 0.0f;
    for (int it = 0; it < 2; ++it) {
        maxpool_sq(mask, H, W, radius, tmp, supp);
        for (size_t i = 0; i < n; ++i) ss[i] = (supp[i] > 0.0f) ? 0.0f : score[i];
        maxpool_sq(ss, H, W, radius, tmp, mp);
        for (size_t i = 0; i < n; ++i)
            if (ss[i] == mp[i] && !(supp[i] > 0.0f)) mask[i] = 1.0f;
    }
    for (size_t i = 0; i < n; ++i) out[i] = mask[i] > 0.0f ? score[i] : 0.0f;
    free(tmp); free(mp); free(mask); free(supp); free(ss);
}

float rfo_sumsq256(const float* x) {
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float s = x[4 * l] * x[4 * l];
        s = fmaf(x[4 * l + 1], x[4 * l + 1], s);
        s = fmaf(x[4 * l + 2], x[4 * l + 2], s);
        s = fmaf(x[4 * l + 3], x[4 * l + 3], s);
        p[l] = s;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float q[64];
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof(p));
    }
    return p[0];
}

void rfo_l2norm256(const float* x, float* y) {
    float d = fmaxf(sqrtf(rfo_sumsq256(x)), 1e-12f);
    for (int c = 0; c < 256; ++c) y[c] = x[c] / d;
}

typedef struct { float s; int32_t idx; } cand_t;
static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int rfo_superpoint(const float* wts, const uint8_t* img, int H, int W, int Kmax, float thr,
                   int nms_radius, int border, int32_t* kxy, float* score, float* desc,
                   float* dbg_scoremap, float* dbg_nms, float* dbg_descmap, float* dbg_feat) {
    return rfo_superpoint_ex(wts, img, H, W, Kmax, thr, nms_radius, border, 0, kxy, score, desc, dbg_scoremap, dbg_nms, dbg_descmap, dbg_feat);
}

int rfo_superpoint_ex(const float* wts, const uint8_t* img, int H, int W, int Kmax, float thr,
                      int nms_radius, int border, int topk_always, int32_t* kxy, float* score, float* desc,
                      float* dbg_scoremap, float* dbg_nms, float* dbg_descmap, float* dbg_feat) {
    /* NormalizeImage: transform.cpp:11  u8 -> f32 * (1/255) */
    size_t n0 = (size_t)H * W;
    float* f = (float*)malloc(sizeof(float) * n0);
    for (size_t i = 0; i < n0; ++i) f[i] = (float)img[i] * 0.003921568859368563f;
    int n = rfo_superpoint_f32(wts, f, H, W, Kmax, thr, nms_radius, border, topk_always, kxy, score, desc, dbg_scoremap, dbg_nms, dbg_descmap, dbg_feat);
    free(f);
    return n;
}

/* The graph proper: what Session::Run sees (superpoint_onnx.cc:105-136) is the float image, whoever normalised it. */
int rfo_superpoint_f32(const float* wts, const float* img, int H, int W, int Kmax, float thr,
                       int nms_radius, int border, int topk_always, int32_t* kxy, float* score, float* desc,
                       float* dbg_scoremap, float* dbg_nms, float* dbg_descmap, float* dbg_feat) {
    /* Any H, W >= 8, like the ONNX graph (dynamic axes): the three 2x2/2 max-pools floor, so the feature grid is Hc x Wc =
     * floor(H/8) x floor(W/8) and everything after the heads lives on the Hs x Ws = 8Hc x 8Wc score map (= the image when H, W
     * are multiples of 8; e.g. KITTI 1241 x 376 -> 155 x 47 cells, score map 1240 x 376). */
    const int Hc = H / 2 / 2 / 2, Wc = W / 2 / 2 / 2;
    const int Himg = H, Wimg = W;
#define WOFF(l) (wts + rfo_sp_layer_offset((l), 0))
#define BOFF(l) (wts + rfo_sp_layer_offset((l), 1))
    size_t n0 = (size_t)H * W;
    float* a = (float*)malloc(sizeof(float) * n0 * 64);
    float* b = (float*)malloc(sizeof(float) * n0 * 64);
    memcpy(a, img, sizeof(float) * n0);
    rfo_conv3x3(a, H, W, 1, WOFF(0), BOFF(0), 64, 1, 0, b);
    rfo_conv3x3(b, H, W, 64, WOFF(1), BOFF(1), 64, 1, 1, a);
    rfo_conv3x3(a, Himg / 2, Wimg / 2, 64, WOFF(2), BOFF(2), 64, 1, 0, b);
    rfo_conv3x3(b, Himg / 2, Wimg / 2, 64, WOFF(3), BOFF(3), 64, 1, 1, a);
    rfo_conv3x3(a, Himg / 2 / 2, Wimg / 2 / 2, 64, WOFF(4), BOFF(4), 128, 1, 0, b);
    rfo_conv3x3(b, Himg / 2 / 2, Wimg / 2 / 2, 128, WOFF(5), BOFF(5), 128, 1, 1, a);
    rfo_conv3x3(a, Hc, Wc, 128, WOFF(6), BOFF(6), 128, 1, 0, b);
    rfo_conv3x3(b, Hc, Wc, 128, WOFF(7), BOFF(7), 128, 1, 0, a); /* a = feat [Hc,Wc,128] */
    if (dbg_feat) memcpy(dbg_feat, a, sizeof(float) * (size_t)Hc * Wc * 128);
    const int cells = Hc * Wc;
    float* pa = (float*)malloc(sizeof(float) * (size_t)cells * 256);
    float* logits = (float*)malloc(sizeof(float) * (size_t)cells * 65);
    rfo_conv3x3(a, Hc, Wc, 128, WOFF(8), BOFF(8), 256, 1, 0, pa);
    rfo_linear(pa, cells, 256, WOFF(9), BOFF(9), 65, logits);
    H = 8 * Hc; W = 8 * Wc;      /* from here on: the score map's frame */
    n0 = (size_t)H * W;
    float* smap = (float*)malloc(sizeof(float) * n0);
    float* nmap = (float*)malloc(sizeof(float) * n0);
    rfo_softmax65_d2s(logits, Hc, Wc, smap);
    if (dbg_scoremap) memcpy(dbg_scoremap, smap, sizeof(float) * n0);
    rfo_nms(smap, H, W, nms_radius, nmap);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if (y < border || y >= H - border || x < border || x >= W - border) nmap[(size_t)y * W + x] = -1.0f;
    if (dbg_nms) memcpy(dbg_nms, nmap, sizeof(float) * n0);
    /* descriptor head */
    float* dmap = (float*)malloc(sizeof(float) * (size_t)cells * 256);
    rfo_conv3x3(a, Hc, Wc, 128, WOFF(10), BOFF(10), 256, 1, 0, pa);
    rfo_linear(pa, cells, 256, WOFF(11), BOFF(11), 256, dmap);
    for (int c = 0; c < cells; ++c) rfo_l2norm256(dmap + (size_t)c * 256, dmap + (size_t)c * 256);
    if (dbg_descmap) memcpy(dbg_descmap, dmap, sizeof(float) * (size_t)cells * 256);
    /* select */
    cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * n0);
    int nc = 0;
    for (size_t i = 0; i < n0; ++i)
        if (nmap[i] > thr) { cand[nc].s = nmap[i]; cand[nc].idx = (int32_t)i; ++nc; }
    int n = nc;
    if (nc > Kmax || topk_always) { qsort(cand, nc, sizeof(cand_t), cand_cmp); n = nc < Kmax ? nc : Kmax; }
    for (int k = 0; k < Kmax; ++k) {
        if (k < n) {
            int x = cand[k].idx % W, y = cand[k].idx / W;
            kxy[2 * k] = x; kxy[2 * k + 1] = y; score[k] = cand[k].s;
            /* sample_descriptors, s = 8, grid_sample(bilinear, align_corners=True, zeros) */
            float gx = (((float)x - 3.5f) / ((float)W - 4.5f)) * 2.0f - 1.0f;
            float gy = (((float)y - 3.5f) / ((float)H - 4.5f)) * 2.0f - 1.0f;
            float ix = ((gx + 1.0f) * 0.5f) * (float)(Wc - 1);
            float iy = ((gy + 1.0f) * 0.5f) * (float)(Hc - 1);
            float fx0 = floorf(ix), fy0 = floorf(iy);
            int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
            float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
            float wnw = wx0 * wy0, wne = wx1 * wy0, wsw = wx0 * wy1, wse = wx1 * wy1;
            float v[256];
            for (int c = 0; c < 256; ++c) {
                float nw = (x0 >= 0 && x0 < Wc && y0 >= 0 && y0 < Hc) ? dmap[((size_t)y0 * Wc + x0) * 256 + c] : 0.0f;
                float ne = (x1 >= 0 && x1 < Wc && y0 >= 0 && y0 < Hc) ? dmap[((size_t)y0 * Wc + x1) * 256 + c] : 0.0f;
                float sw = (x0 >= 0 && x0 < Wc && y1 >= 0 && y1 < Hc) ? dmap[((size_t)y1 * Wc + x0) * 256 + c] : 0.0f;
                float se = (x1 >= 0 && x1 < Wc && y1 >= 0 && y1 < Hc) ? dmap[((size_t)y1 * Wc + x1) * 256 + c] : 0.0f;
                float acc = nw * wnw;
                acc = fmaf(ne, wne, acc);
                acc = fmaf(sw, wsw, acc);
                acc = fmaf(se, wse, acc);
                v[c] = acc;
            }
            rfo_l2norm256(v, desc + (size_t)k * 256);
        } else {
            kxy[2 * k] = 0; kxy[2 * k + 1] = 0; score[k] = 0.0f;
            memset(desc + (size_t)k * 256, 0, sizeof(float) * 256);
        }
    }
    free(a); free(b); free(pa); free(logits); free(smap); free(nmap); free(dmap); free(cand);
    return n;
}

/* ================================================================== */
/* LightGlue                                                           */
/* ================================================================== */
#define LG_D 256
#define LG_H 4
#define LG_HD 64
#define LG_L 9

typedef struct {
    const float *wqkv, *bqkv, *wo, *bo, *w1, *b1, *lng, *lnb, *w2, *b2;               /* self */
    const float *cwqk, *cbqk, *cwv, *cbv, *cwo, *cbo, *cw1, *cb1, *clng, *clnb, *cw2, *cb2; /* cross */
} lg_layer_t;
typedef struct { const float* wr; lg_layer_t L[LG_L]; const float *wp, *bp, *wm, *bm; int64_t total; } lg_w_t;

static void lg_map(const float* w, lg_w_t* o) {
    const float* p = w;
#define TAKE(field, n) do { field = p; p += (n); } while (0)
    TAKE(o->wr, 32 * 2);
    for (int l = 0; l < LG_L; ++l) {
        lg_layer_t* L = &o->L[l];
        TAKE(L->wqkv, 768 * 256); TAKE(L->bqkv, 768); TAKE(L->wo, 256 * 256); TAKE(L->bo, 256);
        TAKE(L->w1, 512 * 512); TAKE(L->b1, 512); TAKE(L->lng, 512); TAKE(L->lnb, 512);
        TAKE(L->w2, 256 * 512); TAKE(L->b2, 256);
        TAKE(L->cwqk, 256 * 256); TAKE(L->cbqk, 256); TAKE(L->cwv, 256 * 256); TAKE(L->cbv, 256);
        TAKE(L->cwo, 256 * 256); TAKE(L->cbo, 256);
        TAKE(L->cw1, 512 * 512); TAKE(L->cb1, 512); TAKE(L->clng, 512); TAKE(L->clnb, 512);
        TAKE(L->cw2, 256 * 512); TAKE(L->cb2, 256);
    }
    TAKE(o->wp, 256 * 256); TAKE(o->bp, 256); TAKE(o->wm, 256); TAKE(o->bm, 1);
    o->total = p - w;
}
int64_t rfo_lg_weight_count(void) { lg_w_t m; lg_map((const float*)0, &m); return m.total; }

/* x + ffn([x | msg]) : Linear(512,512) -> LayerNorm(512) -> GELU(erf) -> Linear(512,256) */
static void lg_ffn_residual(float* x, const float* msg, int n, const float* w1, const float* b1,
                            const float* g, const float* be, const float* w2, const float* b2) {
    float* cat = (float*)malloc(sizeof(float) * (size_t)n * 512);
    float* h = (float*)malloc(sizeof(float) * (size_t)n * 512);
    float* o = (float*)malloc(sizeof(float) * (size_t)n * 256);
    for (int i = 0; i < n; ++i) {
        memcpy(cat + (size_t)i * 512, x + (size_t)i * 256, sizeof(float) * 256);
        memcpy(cat + (size_t)i * 512 + 256, msg + (size_t)i * 256, sizeof(float) * 256);
    }
    rfo_linear(cat, n, 512, w1, b1, 512, h);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float* r = h + (size_t)i * 512;
        float mean = 0.0f;
        for (int c = 0; c < 512; ++c) mean += r[c];
        mean /= 512.0f;
        float var = 0.0f;
        for (int c = 0; c < 512; ++c) { float d = r[c] - mean; var = fmaf(d, d, var); }
        var /= 512.0f;
        float rs = 1.0f / sqrtf(var + 1e-5f);
        for (int c = 0; c < 512; ++c) {
            float v = (r[c] - mean) * rs * g[c] + be[c];
            r[c] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        }
    }
    rfo_linear(h, n, 512, w2, b2, 256, o);
    for (size_t i = 0; i < (size_t)n * 256; ++i) x[i] += o[i];
    free(cat); free(h); free(o);
}

/* softmax(q k^T * scale) v per head; q:[nq,256] k,v:[nk,256] head-major columns */
static void lg_attention(const float* q, const float* k, const float* v, int nq, int nk, float* out) {
#pragma omp parallel
    {
        float* s = (float*)malloc(sizeof(float) * (size_t)nk);
#pragma omp for schedule(static) collapse(2)
        for (int h = 0; h < LG_H; ++h)
            for (int i = 0; i < nq; ++i) {
                const float* qi = q + (size_t)i * 256 + h * 64;
                float m = -INFINITY;
                for (int j = 0; j < nk; ++j) {
                    const float* kj = k + (size_t)j * 256 + h * 64;
                    float acc = 0.0f;
                    for (int d = 0; d < 64; ++d) acc = fmaf(qi[d], kj[d], acc);
                    acc *= 0.125f;
                    s[j] = acc;
                    m = fmaxf(m, acc);
                }
                float sum = 0.0f;
                for (int j = 0; j < nk; ++j) { s[j] = expf(s[j] - m); sum += s[j]; }
                float o[64];
                for (int d = 0; d < 64; ++d) o[d] = 0.0f;
                for (int j = 0; j < nk; ++j) {
                    const float* vj = v + (size_t)j * 256 + h * 64;
                    float pj = s[j];
                    for (int d = 0; d < 64; ++d) o[d] = fmaf(pj, vj[d], o[d]);
                }
                for (int d = 0; d < 64; ++d) out[(size_t)i * 256 + h * 64 + d] = o[d] / sum;
            }
        free(s);
    }
}

/* rotary: t' = t*cos + rot(t)*sin, rot pairs (t[2i],t[2i+1]) -> (-t[2i+1], t[2i]); same table for all heads */
static void lg_rotary(float* t, int n, int stride, const float* cs /*[n,32] cos*/, const float* sn) {
    for (int i = 0; i < n; ++i)
        for (int h = 0; h < LG_H; ++h)
            for (int f = 0; f < 32; ++f) {
                float* p = t + (size_t)i * stride + h * 64 + 2 * f;
                float c = cs[(size_t)i * 32 + f], s = sn[(size_t)i * 32 + f];
                float a = p[0], b = p[1];
                p[0] = a * c - b * s;
                p[1] = b * c + a * s;
            }
}

static void lg_self(float* x, int n, const float* cs, const float* sn, const lg_layer_t* L) {
    float* qkv = (float*)malloc(sizeof(float) * (size_t)n * 768);
    float* q = (float*)malloc(sizeof(float) * (size_t)n * 256);
    float* k = (float*)malloc(sizeof(float) * (size_t)n * 256);
    float* v = (float*)malloc(sizeof(float) * (size_t)n * 256);
    float* ctx = (float*)malloc(sizeof(float) * (size_t)n * 256);
    float* msg = (float*)malloc(sizeof(float) * (size_t)n * 256);
    rfo_linear(x, n, 256, L->wqkv, L->bqkv, 768, qkv);
    for (int i = 0; i < n; ++i) {
        memcpy(q + (size_t)i * 256, qkv + (size_t)i * 768, sizeof(float) * 256);
        memcpy(k + (size_t)i * 256, qkv + (size_t)i * 768 + 256, sizeof(float) * 256);
        memcpy(v + (size_t)i * 256, qkv + (size_t)i * 768 + 512, sizeof(float) * 256);
    }
    lg_rotary(q, n, 256, cs, sn);
    lg_rotary(k, n, 256, cs, sn);
    lg_attention(q, k, v, n, n, ctx);
    rfo_linear(ctx, n, 256, L->wo, L->bo, 256, msg);
    lg_ffn_residual(x, msg, n, L->w1, L->b1, L->lng, L->lnb, L->w2, L->b2);
    free(qkv); free(q); free(k); free(v); free(ctx); free(msg);
}

static void lg_cross(float* x0, float* x1, int M, int N, const lg_layer_t* L) {
    float* qk0 = (float*)malloc(sizeof(float) * (size_t)M * 256);
    float* qk1 = (float*)malloc(sizeof(float) * (size_t)N * 256);
    float* v0 = (float*)malloc(sizeof(float) * (size_t)M * 256);
    float* v1 = (float*)malloc(sizeof(float) * (size_t)N * 256);
    float* c0 = (float*)malloc(sizeof(float) * (size_t)M * 256);
    float* c1 = (float*)malloc(sizeof(float) * (size_t)N * 256);
    float* m0 = (float*)malloc(sizeof(float) * (size_t)M * 256);
    float* m1 = (float*)malloc(sizeof(float) * (size_t)N * 256);
    rfo_linear(x0, M, 256, L->cwqk, L->cbqk, 256, qk0);
    rfo_linear(x1, N, 256, L->cwqk, L->cbqk, 256, qk1);
    rfo_linear(x0, M, 256, L->cwv, L->cbv, 256, v0);
    rfo_linear(x1, N, 256, L->cwv, L->cbv, 256, v1);
    lg_attention(qk0, qk1, v1, M, N, c0); /* m0 = softmax_row(sim) v1   */
    lg_attention(qk1, qk0, v0, N, M, c1); /* m1 = softmax_row(sim^T) v0 */
    rfo_linear(c0, M, 256, L->cwo, L->cbo, 256, m0);
    rfo_linear(c1, N, 256, L->cwo, L->cbo, 256, m1);
    lg_ffn_residual(x0, m0, M, L->cw1, L->cb1, L->clng, L->clnb, L->cw2, L->cb2);
    lg_ffn_residual(x1, m1, N, L->cw1, L->cb1, L->clng, L->clnb, L->cw2, L->cb2);
    free(qk0); free(qk1); free(v0); free(v1); free(c0); free(c1); free(m0); free(m1);
}

static float logsigmoidf_(float z) { /* log(1/(1+exp(-z))) , stable */
    return z >= 0.0f ? -log1pf(expf(-z)) : z - log1pf(expf(z));
}

int rfo_lightglue(const float* weights, const float* k0n, const float* k1n, const float* d0,
                  const float* d1, int M, int N, float filter_thr, int32_t* pairs, float* ms,
                  float* dbg_x0, float* dbg_x1, float* dbg_scores) {
    if (M <= 0 || N <= 0) return 0;
    lg_w_t w; lg_map(weights, &w);
    float* x0 = (float*)malloc(sizeof(float) * (size_t)M * 256);
    float* x1 = (float*)malloc(sizeof(float) * (size_t)N * 256);
    memcpy(x0, d0, sizeof(float) * (size_t)M * 256);
    memcpy(x1, d1, sizeof(float) * (size_t)N * 256);
    float* cs0 = (float*)malloc(sizeof(float) * (size_t)M * 32), *sn0 = (float*)malloc(sizeof(float) * (size_t)M * 32);
    float* cs1 = (float*)malloc(sizeof(float) * (size_t)N * 32), *sn1 = (float*)malloc(sizeof(float) * (size_t)N * 32);
    for (int i = 0; i < M; ++i)
        for (int f = 0; f < 32; ++f) {
            float th = fmaf(w.wr[2 * f + 1], k0n[2 * i + 1], w.wr[2 * f] * k0n[2 * i]);
            cs0[(size_t)i * 32 + f] = cosf(th); sn0[(size_t)i * 32 + f] = sinf(th);
        }
    for (int i = 0; i < N; ++i)
        for (int f = 0; f < 32; ++f) {
            float th = fmaf(w.wr[2 * f + 1], k1n[2 * i + 1], w.wr[2 * f] * k1n[2 * i]);
            cs1[(size_t)i * 32 + f] = cosf(th); sn1[(size_t)i * 32 + f] = sinf(th);
        }
    for (int l = 0; l < LG_L; ++l) {
        lg_self(x0, M, cs0, sn0, &w.L[l]);
        lg_self(x1, N, cs1, sn1, &w.L[l]);
        lg_cross(x0, x1, M, N, &w.L[l]);
    }
    if (dbg_x0) memcpy(dbg_x0, x0, sizeof(float) * (size_t)M * 256);
    if (dbg_x1) memcpy(dbg_x1, x1, sizeof(float) * (size_t)N * 256);
    /* assignment */
    float* md0 = (float*)malloc(sizeof(float) * (size_t)M * 256);
    float* md1 = (float*)malloc(sizeof(float) * (size_t)N * 256);
    rfo_linear(x0, M, 256, w.wp, w.bp, 256, md0);
    rfo_linear(x1, N, 256, w.wp, w.bp, 256, md1);
    for (size_t i = 0; i < (size_t)M * 256; ++i) md0[i] *= 0.25f; /* / 256^(1/4) */
    for (size_t i = 0; i < (size_t)N * 256; ++i) md1[i] *= 0.25f;
    float* sim = (float*)malloc(sizeof(float) * (size_t)M * N);
    rfo_linear(md0, M, 256, md1, NULL, N, sim);
    float* z0 = (float*)malloc(sizeof(float) * M), *z1 = (float*)malloc(sizeof(float) * N);
    rfo_linear(x0, M, 256, w.wm, w.bm, 1, z0);
    rfo_linear(x1, N, 256, w.wm, w.bm, 1, z1);
    float* lse_r = (float*)malloc(sizeof(float) * M), *lse_c = (float*)malloc(sizeof(float) * N);
    for (int i = 0; i < M; ++i) {
        float m = -INFINITY, s = 0.0f;
        for (int j = 0; j < N; ++j) m = fmaxf(m, sim[(size_t)i * N + j]);
        for (int j = 0; j < N; ++j) s += expf(sim[(size_t)i * N + j] - m);
        lse_r[i] = m + logf(s);
    }
    for (int j = 0; j < N; ++j) {
        float m = -INFINITY, s = 0.0f;
        for (int i = 0; i < M; ++i) m = fmaxf(m, sim[(size_t)i * N + j]);
        for (int i = 0; i < M; ++i) s += expf(sim[(size_t)i * N + j] - m);
        lse_c[j] = m + logf(s);
    }
    float* sc = (float*)malloc(sizeof(float) * (size_t)M * N);
    for (int i = 0; i < M; ++i) {
        float li = logsigmoidf_(z0[i]);
        for (int j = 0; j < N; ++j) {
            float sv = sim[(size_t)i * N + j];
            sc[(size_t)i * N + j] = ((sv - lse_r[i]) + (sv - lse_c[j])) + (li + logsigmoidf_(z1[j]));
        }
    }
    if (dbg_scores) memcpy(dbg_scores, sc, sizeof(float) * (size_t)M * N);
    int32_t* a0 = (int32_t*)malloc(sizeof(int32_t) * M), *a1 = (int32_t*)malloc(sizeof(int32_t) * N);
    float* mx0 = (float*)malloc(sizeof(float) * M);
    for (int i = 0; i < M; ++i) {
        float m = -INFINITY; int a = 0;
        for (int j = 0; j < N; ++j) if (sc[(size_t)i * N + j] > m) { m = sc[(size_t)i * N + j]; a = j; }
        a0[i] = a; mx0[i] = m;
    }
    for (int j = 0; j < N; ++j) {
        float m = -INFINITY; int a = 0;
        for (int i = 0; i < M; ++i) if (sc[(size_t)i * N + j] > m) { m = sc[(size_t)i * N + j]; a = i; }
        a1[j] = a;
    }
    int S = 0;
    for (int i = 0; i < M; ++i) {
        if (a1[a0[i]] != i) continue;
        float e = expf(mx0[i]);
        if (e > filter_thr) { pairs[2 * S] = i; pairs[2 * S + 1] = a0[i]; ms[S] = e; ++S; }
    }
    free(x0); free(x1); free(cs0); free(sn0); free(cs1); free(sn1); free(md0); free(md1); free(sim);
    free(z0); free(z1); free(lse_r); free(lse_c); free(sc); free(a0); free(a1); free(mx0);
    return S;
}

void rfo_normalize_keypoints(const float* kxy, int n, int h, int w, float* out) {
    float sx = (float)w / 2, sy = (float)h / 2;
    float scale = (float)MAXI(w, h) / 2;
    for (int i = 0; i < n; ++i) {
        out[2 * i] = (kxy[2 * i] - sx) / scale;
        out[2 * i + 1] = (kxy[2 * i + 1] - sy) / scale;
    }
}

int rfo_postprocess_fused(const int32_t* pairs, const float* ms, int S, float match_thresh,
                          int32_t* vnMatches12, int M) {
    (void)M;
    int size = 0;
    for (int i = 0; i < S; ++i)
        if (ms[i] > match_thresh) { ++size; vnMatches12[pairs[2 * i]] = pairs[2 * i + 1]; }
    return size;
}


/* ================================================================== */
/* Sparse stereo matching: restatement of Frame::ComputeStereoMatches  */
/* (reference src/Frame.cc:1159-1446) for nLevels == 1 (octave 0,      */
/* scale 1), the only configuration the SuperPoint path supports.      */
/* Canonical descriptor distance (DescriptorDistance_sp,               */
/* src/Matchers/SPmatcher.cc:2184-2189 = cv::norm L2): float           */
/* differences, double accumulation; lane l owns dims 4l..4l+3 then an */
/* xor butterfly 32..1 (fixed order shared with the HIP kernel).       */
/* Deviation: keypoints whose 11x11 patch leaves the image rows are    */
/* skipped (the reference's rowRange would throw a cv::Exception).     */
/* ================================================================== */
static float rfo_desc_dist(const float* a, const float* b) {
    double p[64];
    for (int l = 0; l < 64; ++l) {
        double s = 0.0;
        for (int e = 0; e < 4; ++e) { float d = a[4 * l + e] - b[4 * l + e]; s += (double)d * (double)d; }
        p[l] = s;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        double q[64];
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof(p));
    }
    return (float)sqrt(p[0]);
}

typedef struct { int d; int i; } sad_idx_t;
static int sad_idx_cmp(const void* a, const void* b) {
    const sad_idx_t* x = (const sad_idx_t*)a; const sad_idx_t* y = (const sad_idx_t*)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return (x->i > y->i) - (x->i < y->i);
}

void rfo_stereo_match(const uint8_t* imgL, const uint8_t* imgR, int H, int W, const float* kL, int N,
                      const float* kR, int Nr, const float* dL, const float* dR, float mb, float mbf,
                      float* uRight, float* depth) {
    const float TH_HIGH = 1.4f, TH_LOW = 1.2f;
    const float thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const float minD = 0, maxD = mbf / mb;
    sad_idx_t* v = (sad_idx_t*)malloc(sizeof(sad_idx_t) * (size_t)(N > 0 ? N : 1));
    int nv = 0;
    for (int iL = 0; iL < N; ++iL) {
        uRight[iL] = -1.0f; depth[iL] = -1.0f;
        const float uL = kL[2 * iL], vL = kL[2 * iL + 1];
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        float bestDist = TH_HIGH; int bestIdxR = -1;
        for (int iR = 0; iR < Nr; ++iR) {
            const float uR = kR[2 * iR], yR = kR[2 * iR + 1];
            /* row table: right keypoint iR is listed in rows floor(y-2) .. ceil(y+2) (Frame.cc:1207-1218) */
            const int row = (int)vL;
            if (row < (int)floorf(yR - 2.0f) || row > (int)ceilf(yR + 2.0f)) continue;
            if (uR >= minU && uR <= maxU) {
                const float dist = rfo_desc_dist(dL + (size_t)iL * 256, dR + (size_t)iR * 256);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (!(bestDist < thOrbDist) || bestIdxR < 0) continue;
        const float uR0 = kR[2 * bestIdxR];
        const int su = (int)roundf(uL), sv = (int)roundf(vL), sr = (int)roundf(uR0);
        const int w = 5, Lh = 5;
        if (sr - Lh - w < 0 || sr + Lh + w + 1 >= W) continue;                 /* Frame.cc:1325-1330 */
        if (sv - w < 0 || sv + w >= H || su - w < 0 || su + w >= W) continue;   /* see deviation note */
        float vd[11]; float best = 2147483647.0f; int bestinc = 0;
        for (int inc = -Lh; inc <= Lh; ++inc) {
            int sad = 0;
            for (int y = -w; y <= w; ++y)
                for (int x = -w; x <= w; ++x)
                    sad += abs((int)imgL[(size_t)(sv + y) * W + su + x] - (int)imgR[(size_t)(sv + y) * W + sr + inc + x]);
            const float dist = (float)sad;
            if (dist < best) { best = dist; bestinc = inc; }
            vd[Lh + inc] = dist;
        }
        if (bestinc == -Lh || bestinc == Lh) continue;
        const float d1 = vd[Lh + bestinc - 1], d2 = vd[Lh + bestinc], d3 = vd[Lh + bestinc + 1];
        const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = 1.0f * ((float)sr + (float)bestinc + deltaR);
        float disparity = uL - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
            depth[iL] = mbf / disparity;
            uRight[iL] = bestuR;
            v[nv].d = (int)best; v[nv].i = iL; ++nv;
        }
    }
    if (nv > 0) {
        qsort(v, nv, sizeof(sad_idx_t), sad_idx_cmp);
        const float median = (float)v[nv / 2].d;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nv - 1; i >= 0; --i) {
            if ((float)v[i].d < thDist) break;
            uRight[v[i].i] = -1; depth[v[i].i] = -1;
        }
    }
    free(v);
}

/* ================================================================== */
/* the descriptor arithmetic of the classic searches.  */
/* ================================================================== */
/* Best / second-best scan of SearchByProjection1 (src/Matchers/SPmatcher.cc:1218-1248; the same loop  */
/* appears in SearchByProjection :755-800 and Fuse :150-200) for Nq query descriptors against the      */
/* frame's descriptors; candidate lists (CSR: offsets[Nq+1], cand[nnz]) come from the caller's grid    */
/* (Frame::GetFeaturesInArea). skip[f] != 0 stands for "F.mvpMapPoints[f] already has observations".   */
/* nLevels == 1 so every octave is 0. bestDist / bestDist2 start at 256 with strict '<' as there.      */
void rfo_search_candidates(const float* q, int Nq, const float* f, const int32_t* offsets, const int32_t* cand,
                           const uint8_t* skip, int32_t* best_idx, float* best_dist, float* second_dist) {
    for (int i = 0; i < Nq; ++i) {
        float bestDist = 256.f, bestDist2 = 256.f; int bestIdx = -1;
        for (int c = offsets[i]; c < offsets[i + 1]; ++c) {
            const int idx = cand[c];
            if (skip && skip[idx]) continue;
            const float dist = rfo_desc_dist(q + (size_t)i * 256, f + (size_t)idx * 256);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = idx; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist; second_dist[i] = bestDist2;
    }
}

static int float_cmp(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:438-530) over Np map points: desc holds the */
/* observed descriptors of all points back to back, offsets[Np+1] delimits them. Per point: all-pairs   */
/* distances (symmetric, 0 on the diagonal), per row the median sorted[(int)(0.5*(n-1))], first row     */
/* with the strictly smallest median wins. Empty points give best = -1.                                 */
void rfo_distinctive_descriptors(const float* desc, const int32_t* offsets, int Np, int32_t* best, float* median) {
    for (int p = 0; p < Np; ++p) {
        const int o = offsets[p], n = offsets[p + 1] - o;
        if (n <= 0) { best[p] = -1; median[p] = 0.f; continue; }
        float* D = (float*)malloc((size_t)n * n * sizeof(float));
        float* row = (float*)malloc((size_t)n * sizeof(float));
        for (int i = 0; i < n; ++i) {
            D[(size_t)i * n + i] = 0.f;
            for (int j = i + 1; j < n; ++j) {
                const float d = rfo_desc_dist(desc + (size_t)(o + i) * 256, desc + (size_t)(o + j) * 256);
                D[(size_t)i * n + j] = d; D[(size_t)j * n + i] = d;
            }
        }
        float bestMedian = 2147483647.0f; int bestIdx = 0;
        for (int i = 0; i < n; ++i) {
            memcpy(row, D + (size_t)i * n, (size_t)n * sizeof(float));
            qsort(row, n, sizeof(float), float_cmp);
            const float med = row[(int)(0.5 * (n - 1))];
            if (med < bestMedian) { bestMedian = med; bestIdx = i; }
        }
        best[p] = bestIdx; median[p] = bestMedian;
        free(D); free(row);
    }
}

/* ---- threading control for the OpenMP loops above (test infrastructure): a container may expose 256 logical CPUs */
/* with a cgroup quota of 16, and 256 spinning libgomp threads on 16 CPUs run 100x slower than 16 threads.           */
#ifdef _OPENMP
#include <omp.h>
void rfo_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int rfo_get_max_threads(void) { return omp_get_max_threads(); }
#else
void rfo_set_num_threads(int n) { (void)n; }
int rfo_get_max_threads(void) { return 1; }
#endif
