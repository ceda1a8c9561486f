// Crop pre-processing in front of encode_image: box crop -> optional square padding ->
// bicubic resize (Pillow's 8-bit ImagingResample arithmetic: separable, horizontal pass first, 22-bit fixed
// point weights, uint8 rounding after each pass) -> centre crop -> /255, mean/std -> fp32 NCHW.
// Replaces `image.crop(box)` (pre_images/crop_images.py:204-219), `expand2square`
// (utils_tip_cache_and_union_finetune.py:201-212) and `_transform(n_px)` (clipnet/clip.py:75-82).
//
// Integer / byte work, HBM-bound.  Three launches per call:
//   1. preproc_tables_kernel: the fixed-point weight tables (Pillow's precompute_coeffs /
//      normalize_coeffs_8bpc, IEEE double arithmetic with FP contraction off) for the n_px output columns and
//      rows that survive the centre crop - one lane per output index;
//   2. preproc_h_kernel: horizontal pass of the source rows the vertical pass needs -> uint8 scratch;
//   3. preproc_v_kernel: vertical pass, /255, mean/std -> fp32 NCHW (+ optional uint8 NHWC).
//
// Per box: a header of HG_PRE_HDR = 24 int32 words written by the host (hg_api.hip; head[box][...]) and, at word
// offset tab_off[box] of the table buffer, the four arrays filled by preproc_tables_kernel:
//   0 x0  1 y0  2 crop_w  3 crop_h  4 pad_x  5 pad_y  6 src_w  7 src_h  8 ks_h  9 ks_v  10 row_lo  11 n_rows
//   12 tmp_off (bytes)  13 background (0x00BBGGRR)  14 resized_w  15 resized_h  16 left  17 top  (18..23 reserved)
//   h_bounds[n_px][2] (first tap, count)  h_kk[n_px][ks_h]  v_bounds[n_px][2]  v_kk[n_px][ks_v]
// "src" is the virtual image the resize sees: the crop, pasted at (pad_x, pad_y) into src_w x src_h of
// background when square padding is on.  n_rows source rows [row_lo, row_lo + n_rows) feed the vertical pass.
#include "hg_kernels.h"

namespace hg {

static constexpr int PREC = 22;

__device__ __forceinline__ int clip8(int v) {
    v >>= PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__device__ __forceinline__ double cubic(double x) {      // Pillow's bicubic_filter, a = -0.5
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// weight tables: grid (box, axis), one lane per output index of the centre-crop window
__global__ __launch_bounds__(256) void preproc_tables_kernel(const int32_t* __restrict__ head, int32_t* __restrict__ tab,
                                                             const int32_t* __restrict__ tab_off, int n_px) {
#pragma clang fp contract(off)
    const int32_t* t = head + (size_t)blockIdx.x * HG_PRE_HDR;
    int32_t* arr = tab + tab_off[blockIdx.x];
    const int axis = blockIdx.y;                     // 0 horizontal, 1 vertical
    const int in_size = axis ? t[7] : t[6];
    const int out_size = axis ? t[15] : t[14];
    const int first = axis ? t[17] : t[16];
    const int ks = axis ? t[9] : t[8];
    int32_t* bounds = arr + (axis ? 2 * n_px + n_px * t[8] : 0);
    int32_t* kk = bounds + 2 * n_px;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    for (int i = threadIdx.x; i < n_px; i += 256) {
        const double center = ((double)(first + i) + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) ww += cubic(((double)(x + xmin) - center + 0.5) * ss);
        for (int x = 0; x < ks; ++x) {
            double w = 0.0;
            if (x < xmax) {
                w = cubic(((double)(x + xmin) - center + 0.5) * ss);
                if (ww != 0.0) w /= ww;
            }
            kk[(size_t)i * ks + x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PREC)) : (int)(0.5 + w * (double)(1 << PREC));
        }
        bounds[2 * i] = xmin;
        bounds[2 * i + 1] = xmax;
    }
}

// horizontal pass: one workgroup per (source row, box); output tmp[box][row - row_lo][xo][c] (uint8)
__global__ __launch_bounds__(256) void preproc_h_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                        const int32_t* __restrict__ head,
                                                        const int32_t* __restrict__ tab,
                                                        const int32_t* __restrict__ tab_off, int n_px,
                                                        uint8_t* __restrict__ tmp) {
    const int32_t* t = head + (size_t)blockIdx.y * HG_PRE_HDR;
    const int n_rows = t[11];
    const int rr = blockIdx.x;
    if (rr >= n_rows) return;
    const int x0 = t[0], y0 = t[1], cw = t[2], ch = t[3], px = t[4], py = t[5], ks = t[8];
    const int bg = t[13];
    const int32_t* hb = tab + tab_off[blockIdx.y];
    const int32_t* hk = hb + 2 * n_px;
    const int sy = t[10] + rr;                       // row of the virtual source
    const int cy = sy - py;                          // row inside the crop
    const bool row_in_crop = cy >= 0 && cy < ch;
    const int iy = y0 + cy;                          // row of the image
    const bool row_in_img = row_in_crop && iy >= 0 && iy < H;
    uint8_t* dst = tmp + (size_t)(uint32_t)t[12] + (size_t)rr * n_px * 3;
    for (int o = threadIdx.x; o < n_px * 3; o += 256) {
        const int xo = o / 3, c = o - xo * 3;
        const int first = hb[2 * xo], cnt = hb[2 * xo + 1];
        const int32_t* k = hk + (size_t)xo * ks;
        int acc = 1 << (PREC - 1);
        for (int j = 0; j < cnt; ++j) {
            const int cx = first + j - px;           // column inside the crop
            int p;
            if (!row_in_crop || cx < 0 || cx >= cw) p = (bg >> (8 * c)) & 255;      // padding
            else {
                const int ix = x0 + cx;
                p = (row_in_img && ix >= 0 && ix < W) ? img[((size_t)iy * W + ix) * 3 + c] : 0;   // PIL crop: 0 outside
            }
            acc += p * k[j];
        }
        dst[o] = (uint8_t)clip8(acc);
    }
}

// vertical pass + centre crop (already in the tables) + ToTensor/Normalize: one workgroup per (output row, box)
__global__ __launch_bounds__(256) void preproc_v_kernel(const int32_t* __restrict__ head,
                                                        const int32_t* __restrict__ tab,
                                                        const int32_t* __restrict__ tab_off, int n_px,
                                                        const uint8_t* __restrict__ tmp, float* __restrict__ out,
                                                        uint8_t* __restrict__ out_u8, float m0, float m1, float m2,
                                                        float s0, float s1, float s2) {
    const int b = blockIdx.y, yo = blockIdx.x;
    const int32_t* t = head + (size_t)b * HG_PRE_HDR;
    const int ksh = t[8], ksv = t[9], row_lo = t[10];
    const int32_t* vb = tab + tab_off[b] + 2 * n_px + n_px * ksh;
    const int32_t* vk = vb + 2 * n_px;
    const int first = vb[2 * yo] - row_lo, cnt = vb[2 * yo + 1];
    const int32_t* k = vk + (size_t)yo * ksv;
    const uint8_t* src = tmp + (size_t)(uint32_t)t[12];
    // ToTensor + Normalize as a 3 x 256 table (two IEEE divisions per entry instead of per pixel; same bits)
    __shared__ float lut[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) {
        const int c = i >> 8;
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        lut[i] = ((float)(i & 255) / 255.0f - mean) / sd;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_px * 3; i += 256) {
        const int c = i / n_px, xo = i - c * n_px;      // xo fastest: the fp32 plane stores (89 % of the bytes) coalesce
        const int o = xo * 3 + c;
        int acc = 1 << (PREC - 1);
        for (int j = 0; j < cnt; ++j) acc += (int)src[(size_t)(first + j) * n_px * 3 + o] * k[j];
        const int u = clip8(acc);
        if (out_u8) out_u8[((size_t)b * n_px + yo) * n_px * 3 + o] = (uint8_t)u;
        out[(((size_t)b * 3 + c) * n_px + yo) * n_px + xo] = lut[c * 256 + u];
    }
}

hipError_t launch_preprocess(const uint8_t* img, int H, int W, const int32_t* head, int32_t* tab,
                             const int32_t* tab_off, int n, int n_px, int max_rows, uint8_t* tmp, float* out,
                             uint8_t* out_u8, hipStream_t s, bool imagenet_norm) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(preproc_tables_kernel, dim3(n, 2), dim3(256), 0, s, head, tab, tab_off, n_px);
    if (max_rows > 0)
        hipLaunchKernelGGL(preproc_h_kernel, dim3(max_rows, n), dim3(256), 0, s, img, H, W, head, tab, tab_off, n_px, tmp);
    if (imagenet_norm)      // the detector's CLIP view: utils_tip_cache_and_union_finetune.py:86-89
        hipLaunchKernelGGL(preproc_v_kernel, dim3(n_px, n), dim3(256), 0, s, head, tab, tab_off, n_px, tmp, out, out_u8,
                           0.485f, 0.456f, 0.406f, 0.229f, 0.224f, 0.225f);
    else                    // CLIP's own constants: clipnet/clip.py:81
        hipLaunchKernelGGL(preproc_v_kernel, dim3(n_px, n), dim3(256), 0, s, head, tab, tab_off, n_px, tmp, out, out_u8,
                           0.48145466f, 0.4578275f, 0.40821073f, 0.26862954f, 0.26130258f, 0.27577711f);
    return hipGetLastError();
}

}  // namespace hg

// ---- RoI-align over the local feature map ------------------------------------------------
// torchvision.ops.roi_align(feat[1,C,H,W], [boxes], (P,P), spatial_scale, sampling_ratio=-1, aligned=True) and the
// `.flatten(2).mean(-1)` that follows it (upt_tip_cache_model_free_finetune_distill3.py:1026-1037).
// One workgroup per RoI, one lane per channel; the P*P bins are walked in order so that the optional full
// output [n,C,P,P] and the per-RoI mean [n,C] come out of the same pass.  fp32 throughout, in the operation
// order of torchvision's CPU kernel (pre-computed bilinear weights w1..w4 = hy*hx, hy*lx, ly*hx, ly*lx).
namespace hg {

__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ feat, int C, int H, int W,
                                                        const float* __restrict__ boxes, float spatial_scale, int P,
                                                        float* __restrict__ out_pooled, float* __restrict__ out_mean) {
#pragma clang fp contract(off)
    const int n = blockIdx.x;
    const float* b = boxes + 4 * (size_t)n;
    const float sw = b[0] * spatial_scale - 0.5f, sh = b[1] * spatial_scale - 0.5f;
    const float ew = b[2] * spatial_scale - 0.5f, eh = b[3] * spatial_scale - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float count = (float)(gh * gw > 1 ? gh * gw : 1);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float* f = feat + (size_t)c * H * W;
        float total = 0.f;
        for (int ph = 0; ph < P; ++ph)
            for (int pw = 0; pw < P; ++pw) {
                float acc = 0.f;
                for (int iy = 0; iy < gh; ++iy) {
                    float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                    for (int ix = 0; ix < gw; ++ix) {
                        float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                        float yy = y;
                        if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
                        if (yy <= 0.f) yy = 0.f;
                        if (x <= 0.f) x = 0.f;
                        int yl = (int)yy, xl = (int)x, yh, xh;
                        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                        const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                        acc += (hy * hx) * f[yl * W + xl] + (hy * lx) * f[yl * W + xh] + (ly * hx) * f[yh * W + xl] +
                               (ly * lx) * f[yh * W + xh];
                    }
                }
                acc /= count;
                if (out_pooled) out_pooled[(((size_t)n * C + c) * P + ph) * P + pw] = acc;
                total += acc;
            }
        if (out_mean) out_mean[(size_t)n * C + c] = total / (float)(P * P);
    }
}

hipError_t launch_roi_align(const float* feat, int C, int H, int W, const float* boxes, int n, float spatial_scale,
                            int P, float* out_pooled, float* out_mean, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(roi_align_kernel, dim3(n), dim3(256), 0, s, feat, C, H, W, boxes, spatial_scale, P, out_pooled, out_mean);
    return hipGetLastError();
}

}  // namespace hg
