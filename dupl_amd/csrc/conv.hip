// LargeFOV helpers (conv_head.py:11-41): the 3x3 dilated convolutions run as im2col + the MFMA GEMM.
// Activations stay token-major ([pixel][channel]); the column order is (c, tap) so that the conv weight
// (Cout, Cin, 3, 3) is used in place as the GEMM's k-contiguous B operand.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

// x: image b at x + b*img_stride, pixel p at + p*ld, channel c.  col [B*h*w][Cin*9], column = c*9 + tap.
__global__ void im2col_dil3_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int h, int w, int Cin, int dil,
                                   long ld, long img_stride) {
    const long total = (long)B * h * w * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const long r = i / Cin;
        const int px = (int)(r % w), py = (int)((r / w) % h), b = (int)(r / ((long)w * h));
        const float* xb = x + b * img_stride + c;
        float* o = col + r * (long)(Cin * 9) + (long)c * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = py + (t / 3 - 1) * dil, xx = px + (t % 3 - 1) * dil;
            o[t] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? xb[(long)(yy * w + xx) * ld] : 0.f;
        }
    }
}

// adjoint: dx[b][p][c] (+)= sum_tap dcol[b][p - off(tap)][c*9 + tap]
__global__ void col2im_dil3_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int h, int w, int Cin, int dil,
                                   long ld, long img_stride, int accumulate, const float* __restrict__ relu_of) {
    const long total = (long)B * h * w * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const long r = i / Cin;
        const int px = (int)(r % w), py = (int)((r / w) % h), b = (int)(r / ((long)w * h));
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // output pixel q reads input q + off(t); so input p feeds output q = p - off(t)
            const int yy = py - (t / 3 - 1) * dil, xx = px - (t % 3 - 1) * dil;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w)
                s += dcol[((long)b * h * w + (long)yy * w + xx) * (long)(Cin * 9) + (long)c * 9 + t];
        }
        const long off = b * img_stride + (long)(py * w + px) * ld + c;
        if (relu_of && !(relu_of[off] > 0.f)) s = 0.f;   // ReLU backward of the layer that produced x
        float* o = dx + off;
        *o = accumulate ? *o + s : s;
    }
}

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

extern "C" int dupl_im2col_dil3(const float* x, float* col, int32_t B, int32_t h, int32_t w, int32_t Cin, int32_t dil, int64_t ld,
                                int64_t img_stride, dupl_stream_t s) {
    if (!x || !col || B <= 0 || h <= 0 || w <= 0 || Cin <= 0 || dil <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(im2col_dil3_kernel, dim3(ew_grid((long)B * h * w * Cin)), dim3(256), 0, (hipStream_t)s, x, col, B, h, w,
                       Cin, dil, (long)ld, (long)img_stride);
    return dupl_launch_status();
}

extern "C" int dupl_col2im_dil3(const float* dcol, float* dx, int32_t B, int32_t h, int32_t w, int32_t Cin, int32_t dil, int64_t ld,
                                int64_t img_stride, int32_t accumulate, const float* relu_of, dupl_stream_t s) {
    if (!dcol || !dx || B <= 0 || h <= 0 || w <= 0 || Cin <= 0 || dil <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(col2im_dil3_kernel, dim3(ew_grid((long)B * h * w * Cin)), dim3(256), 0, (hipStream_t)s, dcol, dx, B, h, w,
                       Cin, dil, (long)ld, (long)img_stride, accumulate, relu_of);
    return dupl_launch_status();
}
