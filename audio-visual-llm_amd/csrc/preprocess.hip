// Device-side feature extraction: the producer of the hot path's inputs.
//   audio: mono 16 kHz waveform -> Whisper log-mel [80,3000] -> whole-tensor layer norm  (simple_dataset.py:156-186;
//          transformers WhisperFeatureExtractor, models/whisper/feature_extraction_whisper.py:105-133 + audio_utils.spectrogram)
//   video: RGB uint8 frames -> resize(shortest edge, BICUBIC) -> center crop -> rescale -> normalize  (simple_dataset.py:191-264;
//          CLIPImageProcessor PIL backend; Pillow src/libImaging/Resample.c 8-bit fixed point, horizontal pass then vertical)
// The STFT is a 400-point DFT evaluated in float64 (HF's numpy definition promotes to float64; 0.96 GFLOP per clip is noise
// next to the 11.6 TFLOP train step), re/im rounded to float32 as HF stores them, power/mel/log10 in float64, then float32 from
// there on, exactly where HF rounds.  The resize is integer work and bit-exact with Pillow: coefficient tables are computed on
// the host with Pillow's own double-precision recipe and applied by two uint8 passes.
#include "common.h"
#include "avllm_internal.h"
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NMEL_MAX = 128, NSAMP = 480000, NFRAME = 3000, MELW = 32;      // 80 mel bins (Whisper tiny..large-v2) or 128 (large-v3)
constexpr int FR = 8;                        // frames per workgroup

struct LogmelTab {
    double win[NFFT], cs[NFFT], sn[NFFT];
    double w[NMEL_MAX][MELW];
    int lo[NMEL_MAX], cnt[NMEL_MAX];
    int n_mels;
};

__global__ __launch_bounds__(256) void logmel_frames_kernel(const LogmelTab* __restrict__ tab, const float* __restrict__ wave, long ld,
                                                            int n, float* __restrict__ logspec) {
    __shared__ double xw[FR][NFFT];
    __shared__ double cs[NFFT], sn[NFFT];
    __shared__ double pw[FR][NBIN + 3];
    const int b = blockIdx.y, f0 = blockIdx.x * FR, tid = threadIdx.x;
    const int NMEL = tab->n_mels;
    const float* x = wave + (long)b * ld;
    for (int i = tid; i < NFFT; i += 256) { cs[i] = tab->cs[i]; sn[i] = tab->sn[i]; }
    for (int i = tid; i < FR * NFFT; i += 256) {
        const int f = i / NFFT, k = i - f * NFFT;
        int s = (f0 + f) * HOP + k - NFFT / 2;                 // np.pad(..., mode="reflect") of the 30 s zero-padded signal
        if (s < 0) s = -s;
        if (s >= NSAMP) s = 2 * (NSAMP - 1) - s;
        const double v = s < n ? (double)x[s] : 0.0;
        xw[f][k] = v * tab->win[k];
    }
    __syncthreads();
    if (tid < NBIN) {
        double re[FR], im[FR];
#pragma unroll
        for (int f = 0; f < FR; ++f) { re[f] = 0.0; im[f] = 0.0; }
        int idx = 0;
        for (int k = 0; k < NFFT; ++k) {
            const double c = cs[idx], s = sn[idx];
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const double v = xw[f][k];
                re[f] += v * c;
                im[f] -= v * s;
            }
            idx += tid;
            if (idx >= NFFT) idx -= NFFT;
        }
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const double r = (double)(float)re[f], i = (double)(float)im[f];     // stored as complex64 (audio_utils.py:966)
            const double a = sqrt(r * r + i * i);                                  // np.abs(., dtype=float64) ** 2
            pw[f][tid] = a * a;
        }
    }
    __syncthreads();
    for (int o = tid; o < FR * NMEL; o += 256) {
        const int m = o / FR, f = o - m * FR;
        if (f0 + f >= NFRAME) continue;
        const int lo = tab->lo[m], cnt = tab->cnt[m];
        double acc = 0.0;
        for (int j = 0; j < cnt; ++j) acc += tab->w[m][j] * pw[f][lo + j];
        acc = acc > 1e-10 ? acc : 1e-10;
        logspec[((long)b * NMEL + m) * NFRAME + f0 + f] = (float)log10(acc);
    }
}

// per clip: clamp to (max - 8), (x + 4) / 4 in float32 as HF does, then (optionally) F.layer_norm over the whole [80,3000]
__global__ __launch_bounds__(1024) void logmel_finish_kernel(const float* __restrict__ logspec, float* __restrict__ out, int normalize, int NMEL) {
    __shared__ double red[16];
    __shared__ float redf[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long base = (long)blockIdx.x * NMEL * NFRAME;
    const int n = NMEL * NFRAME;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 1024) mx = fmaxf(mx, logspec[base + i]);
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) redf[w] = mx;
    __syncthreads();
    mx = redf[0];
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, redf[i]);
    const float floor_v = mx - 8.0f;
    auto val = [&](int i) { return (fmaxf(logspec[base + i], floor_v) + 4.0f) / 4.0f; };
    if (!normalize) {
        for (int i = tid; i < n; i += 1024) out[base + i] = val(i);
        return;
    }
    auto block_sum = [&](double v) {
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if (lane == 0) red[w] = v;
        __syncthreads();
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        return t;
    };
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) s += (double)val(i);
    const double mean = block_sum(s) / n;
    double q = 0.0;
    for (int i = tid; i < n; i += 1024) { const double d = (double)val(i) - mean; q += d * d; }
    const double rstd = 1.0 / sqrt(block_sum(q) / n + 1e-5);
    for (int i = tid; i < n; i += 1024) out[base + i] = (float)(((double)val(i) - mean) * rstd);
}

// ------------------------------------------------------------------------------------------------- video
constexpr int PBITS = 32 - 8 - 2;            // Resample.c PRECISION_BITS

struct ResizeHdr {
    int H, W, newH, newW, top, left, image, ksx, ksy;
    int off_bx, off_kx, off_by, off_ky, off_lut;          // byte offsets into the plan
};

__device__ __forceinline__ int clip8(int v) {
    v >>= PBITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: in [N,H,W,3] u8 -> tmp [N,H,image,3] u8, only the columns the centre crop keeps
__global__ __launch_bounds__(256) void resize_h_kernel(const char* __restrict__ plan, const unsigned char* __restrict__ in,
                                                       unsigned char* __restrict__ tmp, long total) {
    const ResizeHdr* h = (const ResizeHdr*)plan;
    const int* bx = (const int*)(plan + h->off_bx);
    const int* kx = (const int*)(plan + h->off_kx);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int S = h->image, xo = (int)(i % S);
    const long row = i / S;                                   // n*H + y
    const int x0 = bx[2 * xo], cnt = bx[2 * xo + 1];
    const int* k = kx + xo * h->ksx;
    const unsigned char* p = in + (row * h->W + x0) * 3;
    int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
    for (int j = 0; j < cnt; ++j) {
        const int c = k[j];
        s0 += p[3 * j] * c; s1 += p[3 * j + 1] * c; s2 += p[3 * j + 2] * c;
    }
    unsigned char* o = tmp + i * 3;
    o[0] = (unsigned char)clip8(s0); o[1] = (unsigned char)clip8(s1); o[2] = (unsigned char)clip8(s2);
}

// horizontal pass, staged: a workgroup copies RPB whole input rows (contiguous bytes in [N*H, W*3]) into LDS with aligned dword
// loads, then each thread produces 4 adjacent output pixels (12 bytes, three dword stores) from LDS bytes.  Needs image % 4 == 0.
__global__ __launch_bounds__(256) void resize_h4_kernel(const char* __restrict__ plan, const unsigned char* __restrict__ in,
                                                        unsigned char* __restrict__ tmp, long rows_total, long in_bytes, int rpb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rows_s[];
    const ResizeHdr* h = (const ResizeHdr*)plan;
    const int* bx = (const int*)(plan + h->off_bx);
    const int* kx = (const int*)(plan + h->off_kx);
    const int S = h->image, SV = S >> 2, W3 = h->W * 3, ksx = h->ksx;
    const long row0 = (long)blockIdx.x * rpb;
    const int nrows = (int)(rows_total - row0 < rpb ? rows_total - row0 : rpb);
    const long a0 = row0 * W3, a_al = a0 & ~3L;
    const int head = (int)(a0 - a_al), ndw = (head + nrows * W3 + 3) >> 2;
    for (int i = threadIdx.x; i < ndw; i += 256) {
        const long off = a_al + 4L * i;
        unsigned v;
        if (off + 4 <= in_bytes) v = *(const unsigned*)(in + off);
        else { v = 0; for (int e = 0; e < 4 && off + e < in_bytes; ++e) v |= (unsigned)in[off + e] << (8 * e); }
        ((unsigned*)rows_s)[i] = v;
    }
    __syncthreads();
    const int r = threadIdx.x / SV, xo4 = (threadIdx.x - r * SV) * 4;
    if (r >= nrows) return;
    const unsigned char* src = rows_s + head + r * W3;
    unsigned o[3] = {0u, 0u, 0u};
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        const int xo = xo4 + px, x0 = bx[2 * xo], cnt = bx[2 * xo + 1];
        const int* k = kx + xo * ksx;
        const unsigned char* p = src + x0 * 3;
        int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
        for (int j = 0; j < cnt; ++j) {
            const int c = k[j];
            s0 += p[3 * j] * c; s1 += p[3 * j + 1] * c; s2 += p[3 * j + 2] * c;
        }
        const int b0 = 3 * px;
        o[b0 >> 2] |= (unsigned)clip8(s0) << (8 * (b0 & 3));
        o[(b0 + 1) >> 2] |= (unsigned)clip8(s1) << (8 * ((b0 + 1) & 3));
        o[(b0 + 2) >> 2] |= (unsigned)clip8(s2) << (8 * ((b0 + 2) & 3));
    }
    *(uint3*)(tmp + ((row0 + r) * S + xo4) * 3) = make_uint3(o[0], o[1], o[2]);
}

// vertical pass + rescale/normalize (256-entry table per channel): tmp [N,H,image,3] -> out [N,3,image,image].
// One thread = 4 adjacent output pixels: 12 contiguous tmp bytes per tap (three dword loads), one 16-byte (fp32) or 8-byte (bf16)
// store per channel plane.  VEC=1 is the scalar form for image % 4 != 0.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const char* __restrict__ plan, const unsigned char* __restrict__ tmp,
                                                            T* __restrict__ out, long total) {
    const ResizeHdr* h = (const ResizeHdr*)plan;
    const int* by = (const int*)(plan + h->off_by);
    const int* ky = (const int*)(plan + h->off_ky);
    const float* lut = (const float*)(plan + h->off_lut);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int S = h->image, SV = S / VEC, xo = (int)(i % SV) * VEC, yo = (int)((i / SV) % S);
    const long n = i / ((long)SV * S);
    const int y0 = by[2 * yo], cnt = by[2 * yo + 1];
    const int* k = ky + yo * h->ksy;
    const unsigned char* p = tmp + ((n * h->H + y0) * S + xo) * 3;
    int acc[3 * VEC];
#pragma unroll
    for (int e = 0; e < 3 * VEC; ++e) acc[e] = 1 << (PBITS - 1);
    for (int j = 0; j < cnt; ++j) {
        const int c = k[j];
        const unsigned char* q = p + (long)j * S * 3;
        if (VEC == 4) {
            const uint3 w = *(const uint3*)q;                  // 12 bytes, 4-byte aligned (S % 4 == 0)
            const unsigned d[3] = {w.x, w.y, w.z};
#pragma unroll
            for (int e = 0; e < 12; ++e) acc[e] += (int)((d[e >> 2] >> (8 * (e & 3))) & 0xffu) * c;
        } else {
#pragma unroll
            for (int e = 0; e < 3; ++e) acc[e] += q[e] * c;
        }
    }
    const long plane = (long)S * S, o = n * 3 * plane + (long)yo * S + xo;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float v[VEC];
#pragma unroll
        for (int px = 0; px < VEC; ++px) v[px] = lut[ch * 256 + clip8(acc[3 * px + ch])];
        if constexpr (VEC == 4) store_f<4>(out + o + ch * plane, v);
        else out[o + ch * plane] = (T)v[0];
    }
}

// ---- host: Pillow's coefficient recipe (Resample.c precompute_coeffs + normalize_coeffs_8bpc), bicubic a = -0.5, support 2
double bicubic(double x) {
    const double a = -0.5;
    if (x < 0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
int coeff_ksize(int in_size, int out_size) {
    double fs = (double)in_size / out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)std::ceil(2.0 * fs) * 2 + 1;
}
// coefficients of output samples [o0, o0+cnt_out) of an in_size -> out_size resample
void coeffs(int in_size, int out_size, int o0, int cnt_out, std::vector<int>& bounds, std::vector<int>& kk, int ksize) {
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    bounds.assign((size_t)cnt_out * 2, 0);
    kk.assign((size_t)cnt_out * ksize, 0);
    std::vector<double> w(ksize);
    for (int t = 0; t < cnt_out; ++t) {
        const int xx = o0 + t;
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = bicubic((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            kk[(size_t)t * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PBITS)) : (int)(0.5 + w[x] * (1 << PBITS));
        }
        bounds[2 * t] = xmin; bounds[2 * t + 1] = xmax;
    }
}
void resized_shape(int H, int W, int image, int& nh, int& nw) {     // image_transforms.get_resize_output_image_size(shortest_edge)
    if (W <= H) { nw = image; nh = (int)((double)image * H / W); }
    else        { nh = image; nw = (int)((double)image * W / H); }
}
size_t plan_layout(int H, int W, int image, ResizeHdr& h) {
    int nh, nw;
    resized_shape(H, W, image, nh, nw);
    h.H = H; h.W = W; h.newH = nh; h.newW = nw; h.image = image;
    h.top = (nh - image) / 2; h.left = (nw - image) / 2;
    h.ksx = coeff_ksize(W, nw); h.ksy = coeff_ksize(H, nh);
    size_t off = (sizeof(ResizeHdr) + 15) & ~(size_t)15;
    h.off_bx = (int)off; off += (size_t)image * 2 * 4;
    h.off_kx = (int)off; off += (size_t)image * h.ksx * 4;
    h.off_by = (int)off; off += (size_t)image * 2 * 4;
    h.off_ky = (int)off; off += (size_t)image * h.ksy * 4;
    h.off_lut = (int)off; off += 3 * 256 * 4;
    return (off + 255) & ~(size_t)255;
}

// plans this process initialised: the compute call checks its (H, W, image) against the plan it is handed
std::mutex g_plan_mu;
std::map<const void*, std::array<int, 3>> g_plans;

}  // namespace

// =============================================================================================== C ABI
extern "C" size_t avllm_logmel_table_bytes(void) { return (sizeof(LogmelTab) + 255) & ~(size_t)255; }

extern "C" int avllm_logmel_table_init(void* table_dev, int32_t n_mels) {
    AV_CHECK_ARG(table_dev, "logmel_table_init: null");
    AV_CHECK_ARG(n_mels == 80 || n_mels == 128, "logmel_table_init: n_mels=%d (Whisper uses 80 or 128)", n_mels);
    const int NMEL = n_mels;
    std::vector<char> buf(sizeof(LogmelTab), 0);
    LogmelTab* t = (LogmelTab*)buf.data();
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < NFFT; ++i) {
        t->win[i] = 0.5 + 0.5 * std::cos(pi * (double)(-NFFT + 2 * i) / (double)NFFT);      // np.hanning(401)[:-1]
        t->cs[i] = std::cos(2.0 * pi * i / NFFT);
        t->sn[i] = std::sin(2.0 * pi * i / NFFT);
    }
    // audio_utils.mel_filter_bank(201, 80, 0, 8000, 16000, norm="slaney", mel_scale="slaney")
    auto hz2mel = [](double f) { return f >= 1000.0 ? 15.0 + std::log(f / 1000.0) * (27.0 / std::log(6.4)) : 3.0 * f / 200.0; };
    auto mel2hz = [](double m) { return m >= 15.0 ? 1000.0 * std::exp((std::log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; };
    double ff[NMEL_MAX + 2];
    const double m0 = hz2mel(0.0), m1 = hz2mel(8000.0), step = (m1 - m0) / (NMEL + 1);
    for (int i = 0; i < NMEL + 2; ++i) ff[i] = mel2hz(i == NMEL + 1 ? m1 : m0 + step * i);      // np.linspace end point is exact
    const double fstep = 8000.0 / (NBIN - 1);
    for (int m = 0; m < NMEL; ++m) {
        const double enorm = 2.0 / (ff[m + 2] - ff[m]);
        int lo = -1, cnt = 0;
        for (int k = 0; k < NBIN; ++k) {
            const double fk = k == NBIN - 1 ? 8000.0 : fstep * k;
            const double down = -(ff[m] - fk) / (ff[m + 1] - ff[m]), up = (ff[m + 2] - fk) / (ff[m + 2] - ff[m + 1]);
            double v = down < up ? down : up;
            v = v > 0 ? v : 0;
            if (v > 0) {
                if (lo < 0) lo = k;
                AV_CHECK_ARG(k - lo < MELW, "logmel_table_init: mel filter %d wider than %d bins", m, MELW);
                t->w[m][k - lo] = v * enorm;
                cnt = k - lo + 1;
            }
        }
        t->lo[m] = lo < 0 ? 0 : lo; t->cnt[m] = cnt;
    }
    t->n_mels = NMEL;
    AV_HIP(hipMemcpy(table_dev, buf.data(), sizeof(LogmelTab), hipMemcpyHostToDevice));
    return AV_OK;
}

extern "C" size_t avllm_logmel_workspace_bytes(int32_t B, int32_t n_mels) { return (size_t)B * n_mels * NFRAME * 4 + 256; }

extern "C" int avllm_logmel(const void* table, const float* wave, int32_t B, int32_t n, int64_t ld, int32_t normalize, int32_t n_mels, float* out,
                            void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_CHECK_ARG(table && wave && out && ws && B > 0 && n >= 0 && ld >= n, "logmel: null/empty (B=%d n=%d ld=%ld)", B, n, (long)ld);
    AV_CHECK_ARG(n_mels == 80 || n_mels == 128, "logmel: n_mels=%d must match the table (80 or 128)", n_mels);
    AV_CHECK_ARG(ws_bytes >= avllm_logmel_workspace_bytes(B, n_mels), "logmel: workspace %zu < %zu bytes", ws_bytes, avllm_logmel_workspace_bytes(B, n_mels));
    if (n > NSAMP) n = NSAMP;                                  // truncation=True, max_length = 30 s
    float* logspec = (float*)ws;
    hipLaunchKernelGGL(logmel_frames_kernel, dim3(NFRAME / FR, B), dim3(256), 0, st, (const LogmelTab*)table, wave, (long)ld, n, logspec);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(logmel_finish_kernel, dim3(B), dim3(1024), 0, st, logspec, out, normalize, n_mels);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" size_t avllm_clip_preproc_plan_bytes(int32_t H, int32_t W, int32_t image) {
    if (H <= 0 || W <= 0 || image <= 0) return 0;
    ResizeHdr h;
    return plan_layout(H, W, image, h);
}

extern "C" int avllm_clip_preproc_plan_init(void* plan_dev, int32_t H, int32_t W, int32_t image, const float* mean3, const float* std3) {
    AV_CHECK_ARG(plan_dev && mean3 && std3 && H > 0 && W > 0 && image > 0, "clip_preproc_plan_init: null/empty");
    ResizeHdr h;
    const size_t bytes = plan_layout(H, W, image, h);
    std::vector<char> buf(bytes, 0);
    std::vector<int> b, k;
    coeffs(W, h.newW, h.left, image, b, k, h.ksx);
    memcpy(buf.data() + h.off_bx, b.data(), b.size() * 4); memcpy(buf.data() + h.off_kx, k.data(), k.size() * 4);
    coeffs(H, h.newH, h.top, image, b, k, h.ksy);
    memcpy(buf.data() + h.off_by, b.data(), b.size() * 4); memcpy(buf.data() + h.off_ky, k.data(), k.size() * 4);
    float* lut = (float*)(buf.data() + h.off_lut);
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            const float r = (float)((double)v * (1.0 / 255.0));          // image_transforms.rescale: float64 product, then float32
            lut[c * 256 + v] = (r - mean3[c]) / std3[c];                  // image_transforms.normalize in float32
        }
    memcpy(buf.data(), &h, sizeof(h));
    AV_HIP(hipMemcpy(plan_dev, buf.data(), bytes, hipMemcpyHostToDevice));
    { std::lock_guard<std::mutex> lk(g_plan_mu); g_plans[plan_dev] = {H, W, image}; }
    return AV_OK;
}

extern "C" size_t avllm_clip_preproc_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t image) {
    (void)W;
    return (size_t)N * H * image * 3 + 256;
}

extern "C" int avllm_clip_preproc(const void* plan, const uint8_t* frames, int32_t N, int32_t H, int32_t W, int32_t image, void* out,
                                  int32_t dtype, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_CHECK_ARG(plan && frames && out && ws && N > 0 && H > 0 && W > 0 && image > 0, "clip_preproc: null/empty");
    AV_CHECK_ARG(ws_bytes >= avllm_clip_preproc_workspace_bytes(N, H, W, image), "clip_preproc: workspace %zu too small", ws_bytes);
    AV_CHECK_ARG(dtype == AV_F32 || dtype == AV_BF16, "clip_preproc: dtype %d", dtype);
    AV_CHECK_ARG((long)N * H * W * 3 < (1L << 40), "clip_preproc: batch too large");
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = g_plans.find(plan);
        AV_CHECK_ARG(it != g_plans.end(), "clip_preproc: plan was not initialised with avllm_clip_preproc_plan_init");
        AV_CHECK_ARG(it->second[0] == H && it->second[1] == W && it->second[2] == image, "clip_preproc: plan is for %dx%d->%d, call is %dx%d->%d",
                     it->second[0], it->second[1], it->second[2], H, W, image);
    }
    unsigned char* tmp = (unsigned char*)ws;
    const long rows_total = (long)N * H;
    const int rpb = image % 4 == 0 && image <= 1024 ? 256 / (image / 4) : 0;
    if (rpb > 0 && (size_t)rpb * W * 3 + 8 <= 60 * 1024) {
        const size_t lds = ((size_t)rpb * W * 3 + 8 + 15) & ~(size_t)15;
        hipLaunchKernelGGL(resize_h4_kernel, dim3(av_cdiv(rows_total, rpb)), dim3(256), lds, st, (const char*)plan, frames, tmp, rows_total,
                           rows_total * W * 3, rpb);
    } else {
        const long t1 = rows_total * image;
        hipLaunchKernelGGL(resize_h_kernel, dim3(av_cdiv(t1, 256)), dim3(256), 0, st, (const char*)plan, frames, tmp, t1);
    }
    AV_LAUNCH_CHECK();
    if (image % 4 == 0) {
        const long t2 = (long)N * image * (image / 4);
        if (dtype == AV_F32) hipLaunchKernelGGL((resize_v_norm_kernel<float, 4>), dim3(av_cdiv(t2, 256)), dim3(256), 0, st, (const char*)plan, tmp, (float*)out, t2);
        else hipLaunchKernelGGL((resize_v_norm_kernel<bf16, 4>), dim3(av_cdiv(t2, 256)), dim3(256), 0, st, (const char*)plan, tmp, (bf16*)out, t2);
    } else {
        const long t2 = (long)N * image * image;
        if (dtype == AV_F32) hipLaunchKernelGGL((resize_v_norm_kernel<float, 1>), dim3(av_cdiv(t2, 256)), dim3(256), 0, st, (const char*)plan, tmp, (float*)out, t2);
        else hipLaunchKernelGGL((resize_v_norm_kernel<bf16, 1>), dim3(av_cdiv(t2, 256)), dim3(256), 0, st, (const char*)plan, tmp, (bf16*)out, t2);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}
