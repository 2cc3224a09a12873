// Wav2Lip mel-spectrogram (H1) on gfx950: wav2lip/audio.py:45-51 with wav2lip/hparams.py:33-73.
//
//   y   = lfilter([1, -0.97], [1], wav)                         audio.py:20-23  (float64, as scipy)
//   D   = stft(y, n_fft=800, hop=200, win=800, centred, periodic Hann)           audio.py:57-61
//   S   = 20*log10(max(1e-5, mel_basis(80x401) . |D|)) - 20     audio.py:47,92-105
//   out = clip(8*(S+100)/100 - 4, -4, 4)                        audio.py:110-114
//
// The work is ~0.1 GFLOP per 16640-sample window, so this is one small launch: a workgroup per
// STFT frame, the windowed frame and a 800-entry twiddle table in LDS, a direct real DFT in
// float64 (the reference computes in float64 because lfilter promotes), then the 80 mel dot
// products, log and clip from LDS.  Tables (Hann window, twiddles, Slaney mel basis as librosa
// builds it) are computed once on the host in double and cached on the device.
#include "mf_common.h"
#include <cmath>
#include <vector>
#include <mutex>

namespace {

constexpr int N_FFT = 800, HOP = 200, N_BINS = 401, N_MELS = 80;
constexpr double SR = 16000.0, FMIN = 55.0, FMAX = 7600.0, PREEMPH = 0.97;

struct MelTables {
    double* win = nullptr;     // [800]
    double* tw = nullptr;      // [800][2] cos, sin of 2*pi*k/800
    float* basis = nullptr;    // [80][401] float32 like librosa.filters.mel
    bool ready = false;
};
MelTables g_tab[16];
std::mutex g_mu;

// librosa.core.convert.hz_to_mel / mel_to_hz with htk=False (Slaney's Auditory Toolbox scale)
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

int ensure_tables(int dev) {
    std::lock_guard<std::mutex> lk(g_mu);
    MelTables& t = g_tab[dev];
    if (t.ready) return MF_OK;
    const double PI = 3.14159265358979323846;
    std::vector<double> win(N_FFT), tw(2 * N_FFT);
    for (int j = 0; j < N_FFT; ++j) {
        win[j] = 0.5 - 0.5 * std::cos(2.0 * PI * j / N_FFT);   // scipy get_window('hann', 800, fftbins=True)
        tw[2 * j] = std::cos(2.0 * PI * j / N_FFT);
        tw[2 * j + 1] = std::sin(2.0 * PI * j / N_FFT);
    }
    // librosa.filters.mel(sr, n_fft, n_mels=80, fmin, fmax, htk=False, norm='slaney') -> float32
    std::vector<double> mel_f(N_MELS + 2);
    const double m_lo = hz_to_mel(FMIN), m_hi = hz_to_mel(FMAX);
    for (int i = 0; i < N_MELS + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (N_MELS + 1));
    std::vector<float> basis((size_t)N_MELS * N_BINS);
    for (int i = 0; i < N_MELS; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < N_BINS; ++k) {
            const double fk = (SR / 2.0) * k / (N_BINS - 1);
            const double lower = (fk - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fk) / (mel_f[i + 2] - mel_f[i + 1]);
            // librosa stores the triangle in a float32 array, then scales it in place by enorm
            const float w = (float)std::fmax(0.0, std::fmin(lower, upper));
            basis[(size_t)i * N_BINS + k] = (float)((double)w * enorm);
        }
    }
    MF_HIP(hipMalloc(&t.win, win.size() * sizeof(double)));
    MF_HIP(hipMalloc(&t.tw, tw.size() * sizeof(double)));
    MF_HIP(hipMalloc(&t.basis, basis.size() * sizeof(float)));
    MF_HIP(hipMemcpy(t.win, win.data(), win.size() * sizeof(double), hipMemcpyHostToDevice));
    MF_HIP(hipMemcpy(t.tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    MF_HIP(hipMemcpy(t.basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
    t.ready = true;
    return MF_OK;
}

__global__ __launch_bounds__(256) void k_melspec(const float* __restrict__ wav, int n, int T, int pad_mode,
                                                 const double* __restrict__ win, const double* __restrict__ tw,
                                                 const float* __restrict__ basis, float* __restrict__ out) {
    __shared__ double s_x[N_FFT];
    __shared__ double s_tw[2 * N_FFT];
    __shared__ double s_mag[N_BINS + 7];
    const int t = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < N_FFT; j += 256) {
        int p = t * HOP - N_FFT / 2 + j;   // index into the un-padded pre-emphasised signal
        double y = 0.0;
        bool inside = p >= 0 && p < n;
        if (!inside && pad_mode == 1) {     // np.pad(..., mode='reflect')
            if (p < 0) p = -p;
            if (p >= n) p = 2 * (n - 1) - p;
            inside = p >= 0 && p < n;
        }
        if (inside) y = (double)wav[p] - (p > 0 ? PREEMPH * (double)wav[p - 1] : 0.0);
        s_x[j] = y * win[j];
        s_tw[2 * j] = tw[2 * j];
        s_tw[2 * j + 1] = tw[2 * j + 1];
    }
    __syncthreads();
    for (int f = tid; f < N_BINS; f += 256) {
        double re = 0.0, im = 0.0;
        int k = 0;   // (f * j) mod 800
        for (int j = 0; j < N_FFT; ++j) {
            const double x = s_x[j];
            re = fma(x, s_tw[2 * k], re);
            im = fma(x, s_tw[2 * k + 1], im);
            k += f;
            if (k >= N_FFT) k -= N_FFT;
        }
        s_mag[f] = sqrt(re * re + im * im);
    }
    __syncthreads();
    if (tid < N_MELS) {
        const float* row = basis + (size_t)tid * N_BINS;
        double acc = 0.0;
        for (int k = 0; k < N_BINS; ++k) acc = fma((double)row[k], s_mag[k], acc);
        const double min_level = 1e-5;                            // exp(-100/20*ln 10)
        const double S = 20.0 * log10(fmax(min_level, acc)) - 20.0;
        double v = 8.0 * ((S + 100.0) / 100.0) - 4.0;
        v = fmin(fmax(v, -4.0), 4.0);
        out[(size_t)tid * T + t] = (float)v;
    }
}

}  // namespace

extern "C" int mf_melspec_frames(int n) { return n < 0 ? 0 : 1 + n / HOP; }

extern "C" int mf_melspec(const float* wav, int n, float* out, int pad_mode, void* stream) {
    MF_REQUIRE(n > 0, "melspec: empty signal (n=%d)", n);
    MF_REQUIRE(wav && out, "melspec: null argument");
    MF_REQUIRE(pad_mode == 0 || pad_mode == 1, "melspec: pad_mode must be 0 (zeros) or 1 (reflect)");
    MF_REQUIRE(pad_mode == 0 || n > N_FFT / 2, "melspec: reflect padding needs more than %d samples", N_FFT / 2);
    int dev = 0;
    MF_HIP(hipGetDevice(&dev));
    MF_REQUIRE(dev < 16, "melspec: device index %d not supported", dev);
    int rc = ensure_tables(dev);
    if (rc) return rc;
    const MelTables& t = g_tab[dev];
    const int T = 1 + n / HOP;
    hipLaunchKernelGGL(k_melspec, dim3(T), dim3(256), 0, (hipStream_t)stream, wav, n, T, pad_mode, t.win, t.tw,
                       t.basis, out);
    MF_HIP(hipGetLastError());
    return MF_OK;
}
