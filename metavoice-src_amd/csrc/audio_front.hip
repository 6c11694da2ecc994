// Speaker-reference front end on the GPU (round 5): what `SpeakerEncoder.embed_utterance_from_file` does to the audio
// before the LSTM (fam/quantiser/audio/speaker_encoder/model.py:112-117 and audio.py:10-22, through librosa 0.10.1 -- third party):
//   librosa.load(path, sr=16000)      -> decode (host), mix down, RESAMPLE         : resample_kernel  (polyphase FIR, the filter designed on the host)
//   librosa.effects.trim(y, top_db=20) -> frame RMS in dB against the loudest frame : frame_rms_kernel (+ two comparisons on the host)
//   librosa.feature.melspectrogram     -> |STFT|^2 through a Slaney mel filter bank : mel_kernel
// Until round 4 these were numpy on the host (now oracle/audio_frontend.py, the checker).  All three are HBM-trivial (a 30 s reference is
// 1.9 MB); they are here so that a new voice's first request does not leave the device between the file and the embedding.
//
// Stateless entry points on a private stream; host buffers in, host buffers out (the callers are file loaders).
#include <cstdint>
#include <atomic>
#include <vector>

#include "common.h"
#include "kernels.h"

int mv_fail(int code, const char* fmt, ...);  // engine.hip

namespace {

// y[m] = sum_k h[(m + pre) * down - k * up] x[k]   (scipy.signal.upfirdn's definition; h already carries resample_poly's padding and gain)
__global__ void __launch_bounds__(256) resample_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ h, int taps, int up, int down,
                                                       int64_t pre, int64_t n_out, float* __restrict__ y) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_out) return;
    const int64_t t = (m + pre) * down;            // position on the up-sampled grid
    int64_t k = t / up;                            // newest input sample under the filter
    int j = (int)(t - k * up);                     // its tap
    double acc = 0.0;                              // scipy accumulates in the input's precision; double here, rounded once
    for (; j < taps && k >= 0; j += up, --k)
        if (k < n) acc += (double)h[j] * (double)x[k];
    y[m] = (float)acc;
}

// RMS of the centred frames of librosa.feature.rms (frame_length, hop_length, zero padding of frame_length / 2 on both sides), double
__global__ void __launch_bounds__(256) frame_rms_kernel(const float* __restrict__ x, int64_t n, int frame, int hop, double* __restrict__ rms) {
    __shared__ double red[4];
    const int64_t f = blockIdx.x, start = f * hop - frame / 2;
    double s = 0.0;
    for (int i = threadIdx.x; i < frame; i += 256) {
        const int64_t p = start + i;
        const double v = (p >= 0 && p < n) ? (double)x[p] : 0.0;
        s += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rms[f] = sqrt((red[0] + red[1] + red[2] + red[3]) / (double)frame);
}

// One block per frame: frame f = samples [f hop - n_fft / 2, + n_fft) (zero beyond the signal: center=True, pad_mode="constant"), windowed;
// power spectrum by a direct DFT (n_fft = 400 is not a power of two; 201 bins x 400 samples per frame is nothing), twiddles from a table in
// LDS indexed by (k n) mod n_fft -- exact periodicity, fp32 accumulation in sample order; then the mel filter bank rows.
__global__ void __launch_bounds__(256) mel_kernel(const float* __restrict__ wav, int64_t n, const float* __restrict__ window, const float* __restrict__ fb,
                                                  int n_fft, int hop, int n_mels, float* __restrict__ mel) {
    extern __shared__ float sh[];
    float* xs = sh;                    // [n_fft] windowed frame
    float* cs = xs + n_fft;            // [n_fft] cos(2 pi j / n_fft)
    float* sn = cs + n_fft;            // [n_fft] sin
    float* pw = sn + n_fft;            // [n_fft / 2 + 1] power
    const int64_t f = blockIdx.x, start = f * hop - n_fft / 2;
    const int nb = n_fft / 2 + 1;
    for (int i = threadIdx.x; i < n_fft; i += 256) {
        const int64_t p = start + i;
        xs[i] = ((p >= 0 && p < n) ? wav[p] : 0.f) * window[i];
        float s, c;
        sincospif(2.0f * (float)i / (float)n_fft, &s, &c);
        cs[i] = c;
        sn[i] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += 256) {
        float re = 0.f, im = 0.f;
        int idx = 0;                   // (k i) mod n_fft
        for (int i = 0; i < n_fft; ++i) {
            re = fmaf(xs[i], cs[idx], re);
            im = fmaf(xs[i], sn[idx], im);
            idx += k;
            if (idx >= n_fft) idx -= n_fft;
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    for (int mch = threadIdx.x; mch < n_mels; mch += 256) {
        const float* row = fb + (size_t)mch * nb;
        float s = 0.f;
        for (int k = 0; k < nb; ++k) s = fmaf(pw[k], row[k], s);
        mel[(size_t)f * n_mels + mch] = s;
    }
}

struct Scratch {
    hipStream_t s = nullptr;
    std::vector<void*> bufs;
    ~Scratch() {
        for (void* p : bufs) (void)hipFree(p);
        if (s) (void)hipStreamDestroy(s);
    }
    template <typename T> hipError_t get(T** p, size_t count) {
        hipError_t e = hipMalloc((void**)p, count * sizeof(T) + 64);
        if (e == hipSuccess) bufs.push_back(*p);
        return e;
    }
};
// The device the front end works on: a new host thread's current HIP device is 0, and a request thread of rank r > 0 calling in here would
// otherwise build a context on GPU 0 and run next to rank 0's persistent decode kernel (which needs every CU).  -1: the caller's current device.
std::atomic<int> g_audio_device{-1};
#define AF_DEVICE() do { const int d_ = g_audio_device.load(); if (d_ >= 0) AF_TRY(hipSetDevice(d_)); } while (0)
#define AF_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return mv_fail(MV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

}  // namespace

extern "C" {

int mv_audio_set_device(int32_t device) {
    int n = 0;
    if (device >= 0 && (hipGetDeviceCount(&n) != hipSuccess || device >= n)) return mv_fail(MV_ERR_INVALID, "mv_audio_set_device: no device %d", device);
    g_audio_device.store(device < 0 ? -1 : device);
    return MV_OK;
}

int mv_audio_resample(const float* x, int64_t n, const float* h, int32_t taps, int32_t up, int32_t down, int64_t n_pre_remove, int64_t n_out, float* y) {
    if (!x || !h || !y || n < 1 || taps < 1 || up < 1 || down < 1 || n_out < 1 || n_pre_remove < 0) return mv_fail(MV_ERR_INVALID, "mv_audio_resample: bad argument");
    AF_DEVICE();
    Scratch sc;
    AF_TRY(hipStreamCreateWithFlags(&sc.s, hipStreamNonBlocking));
    float *dx = nullptr, *dh = nullptr, *dy = nullptr;
    AF_TRY(sc.get(&dx, (size_t)n));
    AF_TRY(sc.get(&dh, (size_t)taps));
    AF_TRY(sc.get(&dy, (size_t)n_out));
    AF_TRY(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, sc.s));
    AF_TRY(hipMemcpyAsync(dh, h, (size_t)taps * 4, hipMemcpyHostToDevice, sc.s));
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, sc.s, dx, n, dh, taps, up, down, n_pre_remove, n_out, dy);
    AF_TRY(hipGetLastError());
    AF_TRY(hipMemcpyAsync(y, dy, (size_t)n_out * 4, hipMemcpyDeviceToHost, sc.s));
    AF_TRY(hipStreamSynchronize(sc.s));
    return MV_OK;
}

int mv_audio_frame_rms(const float* x, int64_t n, int32_t frame_length, int32_t hop_length, double* rms_out, int64_t n_frames) {
    if (!x || !rms_out || n < 1 || frame_length < 2 || hop_length < 1 || n_frames < 1) return mv_fail(MV_ERR_INVALID, "mv_audio_frame_rms: bad argument");
    AF_DEVICE();
    if (n_frames != 1 + (n + 2 * (int64_t)(frame_length / 2) - frame_length) / hop_length) return mv_fail(MV_ERR_INVALID, "mv_audio_frame_rms: n_frames must be 1 + (n + 2 (frame / 2) - frame) / hop");
    Scratch sc;
    AF_TRY(hipStreamCreateWithFlags(&sc.s, hipStreamNonBlocking));
    float* dx = nullptr;
    double* dr = nullptr;
    AF_TRY(sc.get(&dx, (size_t)n));
    AF_TRY(sc.get(&dr, (size_t)n_frames));
    AF_TRY(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, sc.s));
    hipLaunchKernelGGL(frame_rms_kernel, dim3((unsigned)n_frames), dim3(256), 0, sc.s, dx, n, frame_length, hop_length, dr);
    AF_TRY(hipGetLastError());
    AF_TRY(hipMemcpyAsync(rms_out, dr, (size_t)n_frames * 8, hipMemcpyDeviceToHost, sc.s));
    AF_TRY(hipStreamSynchronize(sc.s));
    return MV_OK;
}

int mv_audio_mel(const float* wav, int64_t n, const float* window, const float* filterbank, int32_t n_fft, int32_t hop_length, int32_t n_mels,
                 float* mel_out, int64_t n_frames) {
    if (!wav || !window || !filterbank || !mel_out || n < 1 || n_fft < 8 || n_fft > 4096 || (n_fft & 1) || hop_length < 1 || n_mels < 1 || n_mels > 256)
        return mv_fail(MV_ERR_INVALID, "mv_audio_mel: bad argument");
    AF_DEVICE();
    if (n_frames != 1 + n / hop_length) return mv_fail(MV_ERR_INVALID, "mv_audio_mel: n_frames must be 1 + n / hop (centred frames)");
    Scratch sc;
    AF_TRY(hipStreamCreateWithFlags(&sc.s, hipStreamNonBlocking));
    const int nb = n_fft / 2 + 1;
    float *dw = nullptr, *dwin = nullptr, *dfb = nullptr, *dm = nullptr;
    AF_TRY(sc.get(&dw, (size_t)n));
    AF_TRY(sc.get(&dwin, (size_t)n_fft));
    AF_TRY(sc.get(&dfb, (size_t)n_mels * nb));
    AF_TRY(sc.get(&dm, (size_t)n_frames * n_mels));
    AF_TRY(hipMemcpyAsync(dw, wav, (size_t)n * 4, hipMemcpyHostToDevice, sc.s));
    AF_TRY(hipMemcpyAsync(dwin, window, (size_t)n_fft * 4, hipMemcpyHostToDevice, sc.s));
    AF_TRY(hipMemcpyAsync(dfb, filterbank, (size_t)n_mels * nb * 4, hipMemcpyHostToDevice, sc.s));
    const size_t lds = ((size_t)3 * n_fft + nb) * 4;
    hipLaunchKernelGGL(mel_kernel, dim3((unsigned)n_frames), dim3(256), lds, sc.s, dw, n, dwin, dfb, n_fft, hop_length, n_mels, dm);
    AF_TRY(hipGetLastError());
    AF_TRY(hipMemcpyAsync(mel_out, dm, (size_t)n_frames * n_mels * 4, hipMemcpyDeviceToHost, sc.s));
    AF_TRY(hipStreamSynchronize(sc.s));
    return MV_OK;
}

}  // extern "C"
