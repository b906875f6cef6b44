// Fused log-mel front-end for gfx950:  wave -> (dither) -> pre-emphasis -> Hann(320)/512-pt STFT
// -> |.|^2 -> HTK mel(64) -> 10 log10 -> SpecAugment zeros -> per-utterance (x-mean)/std.
// Replaces data_module.py:150-174 (torchaudio MelSpectrogram/AmplitudeToDB, :68-71).
//
// One 64-lane wave owns one frame: the 512-point FFT is three radix-8 Stockham passes, each lane
// holding 8 complex points in registers and exchanging through LDS between passes.  A 256-thread
// workgroup handles a tile of FRAMES_PER_BLOCK consecutive frames of one utterance, so the
// (B,64,T) reference-layout store is done from an LDS transpose in 64-byte runs and the
// channels-last store is a full 256-byte row per wave.
#include "common.h"
#include "ctc_lattice.h"
#include "dropout.h"
#include "mel.h"
#include <math.h>
#include <mutex>

namespace lasr {

static constexpr int kNfft = 512, kWin = 320, kHop = 160, kPad = 32, kMel = 64, kFreq = 257;
static constexpr int kWinOff = (kNfft - kWin) / 2;  // 96
static constexpr int kFramesPerBlock = 16;
static constexpr int kWaves = 4;
static constexpr int kSigLen = kFramesPerBlock * kHop + kNfft;   // 3072 samples staged per block
static constexpr int kMaxBins = 24;   // widest triangular filter of the 64-band HTK bank over 257 bins is 20 bins (checked at init)

static constexpr int kNoiseTab = kFramesPerBlock * kHop + kNfft + 8;   // 3080: the block's samples + the left neighbour, whole groups of 4

// Dither noise generated in the kernel (data_module.py:155 `y += 1e-5 * randn_like(y)`): N(0,1) by Box-Muller from
// Philox4x32-10 keyed by the seed, counter = (sample index / 4, utterance, step): one call serves 4 consecutive samples and
// the values depend on nothing but (seed, step, b, j) - whatever grid computes them (lasr_dither_noise writes the same values).
__device__ __forceinline__ void dither4(unsigned long long seed, unsigned long long step, uint32_t b, uint32_t grp, float (&z)[4]) {
  uint32_t r[4];
  philox4x32_10(grp, b, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed ^ 0x6d656c21u, (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(r[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float rad = sqrtf(-2.0f * __logf(u1));
    const float ang = 6.28318530717958647692f * u2;
    z[2 * h] = rad * __cosf(ang);
    z[2 * h + 1] = rad * __sinf(ang);
  }
}
__device__ __forceinline__ float dither1(unsigned long long seed, unsigned long long step, uint32_t b, int64_t j) {
  float z[4];
  dither4(seed, step, b, (uint32_t)(j >> 2), z);
  const int q = (int)(j & 3);
  return q == 0 ? z[0] : (q == 1 ? z[1] : (q == 2 ? z[2] : z[3]));
}

struct MelTables {
  double window[kWin];
  double tw_re[kNfft];
  double tw_im[kNfft];
  float fb[kFreq * kMel];  // [k][m]
  int lo[kMel];            // first bin with non-zero weight
  int hi[kMel];            // last bin with non-zero weight
};
__device__ MelTables g_mel;

static int init_tables() {
  static std::once_flag once;
  static int rc = 0;
  std::call_once(once, [] {
    MelTables* t = new MelTables();
    // the reference window is the f32 periodic Hann torch.hann_window returns
    for (int n = 0; n < kWin; ++n) t->window[n] = (double)(float)(0.5 - 0.5 * cos(2.0 * M_PI * n / kWin));
    for (int n = 0; n < kNfft; ++n) {
      t->tw_re[n] = cos(-2.0 * M_PI * n / kNfft);
      t->tw_im[n] = sin(-2.0 * M_PI * n / kNfft);
    }
    // HTK mel filterbank, torchaudio 0.8.1 create_fb_matrix(257, 0, 8000, 64, 16000, norm=None), f32 steps
    float f_pts[kMel + 2];
    const float m_min = 0.f, m_max = 2595.0f * log10f(1.0f + 8000.0f / 700.0f);
    for (int i = 0; i < kMel + 2; ++i) {
      float m = m_min + (m_max - m_min) * (float)i / (float)(kMel + 1);
      f_pts[i] = 700.0f * (powf(10.0f, m / 2595.0f) - 1.0f);
    }
    for (int m = 0; m < kMel; ++m) { t->lo[m] = kFreq; t->hi[m] = -1; }
    for (int k = 0; k < kFreq; ++k) {
      const float f = 8000.0f * (float)k / (float)(kFreq - 1);
      for (int m = 0; m < kMel; ++m) {
        const float down = (f - f_pts[m]) / (f_pts[m + 1] - f_pts[m]);
        const float up = (f_pts[m + 2] - f) / (f_pts[m + 2] - f_pts[m + 1]);
        const float w = fmaxf(0.f, fminf(down, up));
        t->fb[k * kMel + m] = w;
        if (w > 0.f) {
          if (k < t->lo[m]) t->lo[m] = k;
          if (k > t->hi[m]) t->hi[m] = k;
        }
      }
    }
    for (int m = 0; m < kMel; ++m)
      if (t->hi[m] - t->lo[m] + 1 > kMaxBins) rc = fail(LASR_E_SHAPE, "mel filter %d spans %d bins (> %d)", m, t->hi[m] - t->lo[m] + 1, kMaxBins);
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_mel), t, sizeof(MelTables));
    delete t;
    if (e != hipSuccess) rc = hip_fail(e, "mel table upload");
  });
  return rc;
}

// The transform runs in f64: the kernel is bound by its 20 MB of HBM traffic, not by 1.5 GFLOP of
// butterflies, and f64 removes FFT round-off from the weak bins of high-dynamic-range frames.
struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ void bfly(cplx& a, cplx& b) {
  cplx t = a;
  a = {t.re + b.re, t.im + b.im};
  b = {t.re - b.re, t.im - b.im};
}
__device__ __forceinline__ cplx mul_negi(cplx a) { return {a.im, -a.re}; }  // a * (-i)

// 8-point DFT in registers (decimation in frequency).  Natural-order result k lives in v[kPerm[k]].
__device__ __forceinline__ void fft8(cplx (&v)[8]) {
  const double h = 0.70710678118654752440;
  bfly(v[0], v[4]); bfly(v[1], v[5]); bfly(v[2], v[6]); bfly(v[3], v[7]);
  v[5] = cmul(v[5], cplx{h, -h});
  v[6] = mul_negi(v[6]);
  v[7] = cmul(v[7], cplx{-h, -h});
  bfly(v[0], v[2]); bfly(v[1], v[3]); v[3] = mul_negi(v[3]);
  bfly(v[0], v[1]); bfly(v[2], v[3]);
  bfly(v[4], v[6]); bfly(v[5], v[7]); v[7] = mul_negi(v[7]);
  bfly(v[4], v[5]); bfly(v[6], v[7]);
}
__device__ static const int kPerm[8] = {0, 4, 2, 6, 1, 5, 3, 7};

// Each wave transforms its own frame in its own LDS rows: the exchanges between FFT passes only need the wave's
// own LDS traffic ordered (LDS executes a wave's accesses in order), not a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// grid: (ceil(T/16), B), block 256.  dB values -> db_out [B][T][64] f32 (workspace); per-block
// (sum, sumsq) in double -> partials[b][blk][2].
// Exchange rows of the FFT passes: element n of a wave's row sits at n + (n / 64) * kExPad doubles (mx()).  Pass 1 writes lane (g, k) =
// (lane / 8, lane % 8) at n = 64 g + k + 8 r: with a 512-byte block per g the eight g's of a wave hit the same banks (8-way conflicts
// on 16 of the 32 LDS instructions of a frame pair; 46 % of the kernel's LDS cycles were conflicts); 8 doubles of padding per block
// rotate consecutive g's by 16 banks - two accesses per bank per half-wave, which 64-bit accesses cannot beat.  The unit-stride
// accesses of the other passes keep their pattern (round 5).
static constexpr int kExPad = 8;
static constexpr int kExRow = kNfft + (kNfft / 64 - 1) * kExPad + 8;       // 576 doubles
__device__ __forceinline__ int mx(int n) { return n + (n >> 6) * kExPad; }
#ifdef LASR_MEL_STAMPS
__device__ unsigned long long* g_mel_stamps = nullptr;   // debug builds only (tools/mel_stamps.py): phase times of every workgroup's wave 0
#define MEL_STAMP(i_) do { if (g_mel_stamps && threadIdx.x == 0) g_mel_stamps[((int64_t)by * nbx + bx) * 8 + (i_)] = wall_clock64(); } while (0)
#else
#define MEL_STAMP(i_) do {} while (0)
#endif
struct MelSmem {
  double re[kWaves][kExRow];
  double im[kWaves][kExRow];
  double twr[kNfft];
  double twi[kNfft];
  double red[kWaves][2];
  double win[kWin];
  float sig[kSigLen];
};   // 55 872 bytes
// (bx, by) of (nbx, B): frame tile and utterance - the kernel's own grid, or a slice of the fused feature + lattice grid
__device__ __forceinline__ void mel_db_body(const WaveSrc src, const int32_t* __restrict__ sample_lens,
                                            const int32_t* __restrict__ aug,
                                            int64_t L, int64_t T, float* __restrict__ db_out,
                                            double* __restrict__ partials, int32_t* __restrict__ frames_out,
                                            float* __restrict__ pct_out, int bx, int by, int nbx, MelSmem& sm) {
  double (&s_re)[kWaves][kExRow] = sm.re;
  double (&s_im)[kWaves][kExRow] = sm.im;
  double (&s_twr)[kNfft] = sm.twr;
  double (&s_twi)[kNfft] = sm.twi;
  double (&s_red)[kWaves][2] = sm.red;
  double (&s_win)[kWin] = sm.win;
  float (&s_sig)[kSigLen] = sm.sig;

  const int b = by;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  MEL_STAMP(0);
  // sample_lens[b]: valid samples; bit LASR_LEN_LEAD set = the row starts with ONE lead-in sample x[-1] in front of them (a crop
  // that does not begin at the file's first sample: the reference pre-emphasises BEFORE it crops, data_module.py:157-159, so the
  // crop's first sample is y[loc] - 0.97 y[loc-1], dither of both included)
  const int32_t lw = sample_lens ? sample_lens[b] : 0;
  const int off = (lw >> 30) & 1;
  const int64_t Lb = sample_lens ? (int64_t)(lw & (LASR_LEN_LEAD - 1)) : L;
  const int64_t Lp = Lb + 2 * kPad;
  const int64_t Tb = 1 + Lp / kHop;
  const int64_t pitch = src.pitch ? src.pitch : L;
  const float* y = reinterpret_cast<const float*>(src.wave) + (int64_t)b * pitch;
  const int16_t* y16 = reinterpret_cast<const int16_t*>(src.wave) + (int64_t)b * pitch;
  const float* nz = src.dither ? src.dither + (int64_t)b * pitch : nullptr;
  const bool gen = !nz && src.dstep != nullptr;      // noise generated here (workgroup-uniform)
  for (int i = threadIdx.x; i < kNfft; i += 256) { s_twr[i] = g_mel.tw_re[i]; s_twi[i] = g_mel.tw_im[i]; }
  for (int i = threadIdx.x; i < kWin; i += 256) s_win[i] = g_mel.window[i];
  // this lane's mel filter (lane = mel channel): its <= kMaxBins weights live in registers for all of the block's
  // frames.  (Fetching them inside the per-frame loop, behind its data-dependent bounds, was a chain of
  // dependent global loads per frame.)
  const int fb_lo = g_mel.lo[lane], fb_hi = g_mel.hi[lane];
  float fbw[kMaxBins];
#pragma unroll
  for (int k = 0; k < kMaxBins; ++k) {
    const float w = g_mel.fb[min(fb_lo + k, kFreq - 1) * kMel + lane];
    fbw[k] = (fb_lo + k <= fb_hi) ? w : 0.f;
  }
  if (bx == 0 && threadIdx.x == 0) {
    frames_out[b] = (int32_t)Tb;
    pct_out[b] = (float)Tb / (float)T;
  }
  int rx = 0, wx = 0, ry = 0, wy = 0;
  if (aug) { rx = aug[b * 4 + 0]; wx = aug[b * 4 + 1]; ry = aug[b * 4 + 2]; wy = aug[b * 4 + 3]; }
  // The block's stretch of the reflect-padded, zero-padded, pre-emphasised, dithered signal -> LDS once
  // (16 frames x hop 160 + one window): every load unconditional (clamped index, masked value) and all of a
  // thread's loads issued before the first is used.  Loading per frame behind per-lane bounds branches made
  // each frame a chain of ~6 exposed memory round trips (the kernel took 136 us for 20 MB).
  {
    constexpr int kSigIt = kSigLen / 256;          // 12
    const int64_t i0 = (int64_t)bx * kFramesPerBlock * kHop - kNfft / 2;   // padded-signal index of s_sig[0]
    // generated dither: the noise of utterance samples [jbase, jbase + kNoiseTab) goes through LDS first (one Philox call
    // per 4 samples; the FFT rows are not in use yet), the few reflected samples outside that stretch are computed directly
    float* s_nz = reinterpret_cast<float*>(&sm.re[0][0]);
    const int64_t jbase = i0 - kPad - 4;           // a multiple of 4
    const unsigned long long dstep = gen ? *src.dstep : 0ull;
    if (gen) {
      for (int g = threadIdx.x; g < kNoiseTab / 4; g += 256) {
        float z[4];
        dither4(src.dseed, dstep, (uint32_t)b, (uint32_t)((jbase >> 2) + g), z);
        *reinterpret_cast<float4*>(s_nz + 4 * g) = make_float4(z[0], z[1], z[2], z[3]);
      }
      __syncthreads();
    }
    float cur[kSigIt], prv[kSigIt], ncur[kSigIt], nprv[kSigIt];
    int64_t jcs[kSigIt], jps[kSigIt];
#pragma unroll
    for (int u = 0; u < kSigIt; ++u) {
      int64_t i = i0 + threadIdx.x + 256 * u;
      i = i < 0 ? -i : i;                                   // reflect at both ends (center=True, pad_mode reflect)
      i = i >= Lp ? 2 * (Lp - 1) - i : i;
      const int64_t j = i - kPad;                           // index into the utterance
      const int64_t jmax = max(Lb - 1, (int64_t)0);
      jcs[u] = min(max(j, (int64_t)0), jmax) + off;                 // row indices: the signal sits `off` samples into the row
      jps[u] = min(max(j - 1 + off, (int64_t)0), jmax + off);
    }
    if (src.pcm16) {   // 16-bit PCM as stored in the wav file: /32768 is exact in f32 (what torchaudio.load's normalisation yields)
      int16_t c16[kSigIt], p16[kSigIt];
#pragma unroll
      for (int u = 0; u < kSigIt; ++u) { c16[u] = y16[jcs[u]]; p16[u] = y16[jps[u]]; }
#pragma unroll
      for (int u = 0; u < kSigIt; ++u) { cur[u] = (float)c16[u] * (1.0f / 32768.0f); prv[u] = (float)p16[u] * (1.0f / 32768.0f); }
    } else {
#pragma unroll
      for (int u = 0; u < kSigIt; ++u) { cur[u] = y[jcs[u]]; prv[u] = y[jps[u]]; }
    }
    if (nz) {
#pragma unroll
      for (int u = 0; u < kSigIt; ++u) { ncur[u] = nz[jcs[u]]; nprv[u] = nz[jps[u]]; }
    } else if (gen) {
#pragma unroll
      for (int u = 0; u < kSigIt; ++u) {
        const int64_t dc = jcs[u] - jbase, dp = jps[u] - jbase;
        ncur[u] = (dc >= 0 && dc < kNoiseTab) ? s_nz[dc] : dither1(src.dseed, dstep, (uint32_t)b, jcs[u]);
        nprv[u] = (dp >= 0 && dp < kNoiseTab) ? s_nz[dp] : dither1(src.dseed, dstep, (uint32_t)b, jps[u]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < kSigIt; ++u) { ncur[u] = 0.f; nprv[u] = 0.f; }
    }
    const bool noisy = nz || gen;
#pragma unroll
    for (int u = 0; u < kSigIt; ++u) {
      int64_t i = i0 + threadIdx.x + 256 * u;
      i = i < 0 ? -i : i;
      i = i >= Lp ? 2 * (Lp - 1) - i : i;
      const int64_t j = i - kPad;
      // f32 steps, as the reference computes them (data_module.py:155,157)
      float c = cur[u], p = prv[u];
      if (noisy) { c += 1e-5f * ncur[u]; p += 1e-5f * nprv[u]; }
      const float v = (j == 0 && !off) ? c : c - 0.97f * p;
      s_sig[threadIdx.x + 256 * u] = (j < 0 || j >= Lb || i < 0) ? 0.f : v;
    }
  }
  __syncthreads();
  MEL_STAMP(1);

  double acc_s = 0.0, acc_q = 0.0;
  double* sre = s_re[wid];
  double* sim = s_im[wid];
  // Two real frames per complex transform: frame A in the real part, frame B in the imaginary part,
  //   X_A[k] = (Z[k] + conj(Z[N-k])) / 2,   X_B[k] = (Z[k] - conj(Z[N-k])) / (2i)
  // - half the butterflies and LDS exchanges per frame (the kernel is bound by LDS traffic, not by HBM).  In f64 the
  // cross-talk between the two frames is ~1e-16 of the louder one.
  for (int it = 0; it < kFramesPerBlock / kWaves; it += 2) {
    const int flA = it * kWaves + wid, flB = (it + 1) * kWaves + wid;         // frames within the block
    const int64_t fA = (int64_t)bx * kFramesPerBlock + flA, fB = (int64_t)bx * kFramesPerBlock + flB;
    const bool liveA = fA < Tb && fA < T, liveB = fB < Tb && fB < T;          // wave-uniform
    cplx v[8];
    // ---- pass 0 (Ns = 1): lane j loads x[j + 64 r]; only n in [96, 416) is inside the window
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int n = lane + 64 * r;
      const bool inw = n >= kWinOff && n < kWinOff + kWin;
      const double wn = s_win[min(max(n - kWinOff, 0), kWin - 1)];
      const double xa = (inw && liveA) ? (double)s_sig[flA * kHop + n] * wn : 0.0;
      const double xb = (inw && liveB) ? (double)s_sig[flB * kHop + n] * wn : 0.0;
      v[r] = {xa, xb};
    }
    fft8(v);
    wave_sync();  // previous iteration's readers are done with sre/sim
#pragma unroll
    for (int r = 0; r < 8; ++r) { sre[mx(lane * 8) + r] = v[kPerm[r]].re; sim[mx(lane * 8) + r] = v[kPerm[r]].im; }
    wave_sync();
    if (it == 0) MEL_STAMP(2);
    // ---- pass 1 (Ns = 8)
    {
      const int k = lane & 7;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        cplx x = {sre[mx(lane + 64 * r)], sim[mx(lane + 64 * r)]};
        const int tw = r * k * 8;
        v[r] = cmul(x, cplx{s_twr[tw], s_twi[tw]});
      }
      fft8(v);
      wave_sync();
      const int base = mx((lane >> 3) * 64) + k;             // (k + 8 r < 64: inside the block)
#pragma unroll
      for (int r = 0; r < 8; ++r) { sre[base + r * 8] = v[kPerm[r]].re; sim[base + r * 8] = v[kPerm[r]].im; }
      wave_sync();
    }
    if (it == 0) MEL_STAMP(3);
    // ---- pass 2 (Ns = 64): Z[lane + 64 r] = v[perm r]
    {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        cplx x = {sre[mx(lane + 64 * r)], sim[mx(lane + 64 * r)]};
        const int tw = r * lane;
        v[r] = cmul(x, cplx{s_twr[tw], s_twi[tw]});
      }
      fft8(v);
      wave_sync();
#pragma unroll
      for (int r = 0; r < 8; ++r) { sre[mx(lane + 64 * r)] = v[kPerm[r]].re; sim[mx(lane + 64 * r)] = v[kPerm[r]].im; }
      wave_sync();
    }
    if (it == 0) MEL_STAMP(4);
    // ---- split the two spectra: powers of bins 0..256 of A and B (bin k pairs with N - k)
    double pa[5], pb[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int k = (r < 4) ? lane + 64 * r : 256;
      const int km = (kNfft - k) & (kNfft - 1);
      const double zr = sre[mx(k)], zi = sim[mx(k)], mr = sre[mx(km)], mi = sim[mx(km)];
      const double ar = zr + mr, ai = zi - mi, br = zi + mi, bi = mr - zr;
      pa[r] = 0.25 * (ar * ar + ai * ai);
      pb[r] = 0.25 * (br * br + bi * bi);
    }
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) { sre[lane + 64 * r] = pa[r]; sim[lane + 64 * r] = pb[r]; }
    if (lane == 0) { sre[256] = pa[4]; sim[256] = pb[4]; }
    wave_sync();
    if (it == 0) MEL_STAMP(5);
    // ---- mel + dB: lane = mel channel, both frames
    double mA = 0.0, mB = 0.0;
#pragma unroll
    for (int k = 0; k < kMaxBins; ++k) {   // ascending bins, zero weights past hi
      const int bin = min(fb_lo + k, kFreq - 1);
      mA = fma(sre[bin], (double)fbw[k], mA);
      mB = fma(sim[bin], (double)fbw[k], mB);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool live = h ? liveB : liveA;
      const int64_t f = h ? fB : fA;
      float db = 0.f;
      if (live) {
        db = (float)(10.0 * log10(fmax(h ? mB : mA, 1e-10)));
        if (aug && ((lane >= rx && lane < rx + wx) || (f >= ry && f < ry + wy))) db = 0.f;
        acc_s += (double)db;
        acc_q += (double)db * (double)db;
      }
      if (f < T) db_out[((int64_t)b * T + f) * kMel + lane] = db;
    }
    if (it == 0) MEL_STAMP(6);
  }
  MEL_STAMP(7);
  acc_s = wave_sum_d(acc_s);
  acc_q = wave_sum_d(acc_q);
  if (lane == 0) { s_red[wid][0] = acc_s; s_red[wid][1] = acc_q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0, q = 0;
    for (int w = 0; w < kWaves; ++w) { s += s_red[w][0]; q += s_red[w][1]; }
    partials[((int64_t)b * nbx + bx) * 2 + 0] = s;
    partials[((int64_t)b * nbx + bx) * 2 + 1] = q;
  }
}

__global__ __launch_bounds__(256) void mel_db_kernel(const WaveSrc src, const int32_t* __restrict__ sample_lens,
                                                     const int32_t* __restrict__ aug,
                                                     int64_t L, int64_t T, float* __restrict__ db_out,
                                                     double* __restrict__ partials, int32_t* __restrict__ frames_out,
                                                     float* __restrict__ pct_out) {
  __shared__ MelSmem sm;
  mel_db_body(src, sample_lens, aug, L, T, db_out, partials, frames_out, pct_out, blockIdx.x, blockIdx.y, gridDim.x, sm);
}

// The CTC lattice (one workgroup of two busy waves per utterance, ~0.1 ms of dependent steps) and the log-mel transform
// of the NEXT batch in ONE grid: the 32 lattice workgroups come first and run for the whole launch, the 2 016 feature
// workgroups fill the other 224 CUs meanwhile.  Two queues cannot do this (measured: the second queue's kernels start
// ~1.7 ms after the event they wait for); one grid can.
struct MelCtcArgs {
  // lattice
  const float* logp; const int64_t* targets; const int32_t* in_lens; const int32_t* tgt_lens;
  int64_t T, C, S_max; int blank; float* alpha; float* beta; int32_t* next_same; float* nll; int n_ctc;
  // features
  WaveSrc src; const int32_t* sample_lens; const int32_t* aug;
  int64_t L, Tm; float* db_out; double* partials; int32_t* frames_out; float* pct_out; int nbx;
};
// COMPACT: the lattice of the large-vocabulary head (ctc_lean.hip): `logp` is the gathered emission matrix E [N][C = S_max + 1 ...]
template <int NS, bool COMPACT = false>
__global__ __launch_bounds__(256) void mel_ctc_kernel(MelCtcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if ((int)blockIdx.x < a.n_ctc) {                       // workgroup-uniform
    int32_t* s_tg = reinterpret_cast<int32_t*>(smem_raw);
    float* s_lp = reinterpret_cast<float*>(smem_raw + kCtcMaxS * sizeof(int32_t));
    ctc_alpha_beta_body<NS, true, 256, COMPACT>(a.logp, a.targets, a.in_lens, a.tgt_lens, a.T, a.C, a.S_max, a.blank, a.alpha, a.beta, a.next_same,
                                                a.nll, (int)blockIdx.x, s_tg, s_lp);
  } else {
    const int id = (int)blockIdx.x - a.n_ctc;
    mel_db_body(a.src, a.sample_lens, a.aug, a.L, a.Tm, a.db_out, a.partials, a.frames_out, a.pct_out, id % a.nbx, id / a.nbx,
                a.nbx, *reinterpret_cast<MelSmem*>(smem_raw));
  }
}

// grid: (ceil(T/16), B), block 256: normalise a [16 frames][64] tile, write both layouts.
template <typename T>
__global__ __launch_bounds__(256) void mel_norm_kernel(const float* __restrict__ db, const double* __restrict__ partials,
                                                       const int32_t* __restrict__ frames, int64_t Tt, int nblk,
                                                       int normalize, float* __restrict__ out_bft, T* __restrict__ out_btf,
                                                       unsigned long long* __restrict__ dither_step) {
  __shared__ float tile[kFramesPerBlock][kMel + 1];
  // generated dither: this batch's transform is done (the kernel runs behind it), the next call draws fresh noise
  if (dither_step && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *dither_step += 1ull;
  __shared__ float s_mu, s_rstd;
  const int b = blockIdx.y;
  const int64_t Tb = frames[b];
  if (threadIdx.x < 64) {
    double s = 0, q = 0;
    for (int i = threadIdx.x; i < nblk; i += 64) {
      s += partials[((int64_t)b * nblk + i) * 2 + 0];
      q += partials[((int64_t)b * nblk + i) * 2 + 1];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (threadIdx.x == 0) {
      const double n = (double)Tb * kMel;
      const double mu = s / n;
      double var = (q - n * mu * mu) / (n - 1.0);  // unbiased, torch.std_mean default
      if (var < 0) var = 0;
      s_mu = normalize ? (float)mu : 0.f;
      // (a constant spectrogram - an empty clip, which the reference's reflect pad refuses - normalises to zeros, not to 0/0)
      s_rstd = normalize ? (var > 0 ? (float)(1.0 / sqrt(var)) : 0.f) : 1.f;
    }
  }
  __syncthreads();
  const float mu = s_mu, rstd = s_rstd;
  const int64_t f0 = (int64_t)blockIdx.x * kFramesPerBlock;
  for (int i = threadIdx.x; i < kFramesPerBlock * kMel; i += 256) {
    const int fr = i >> 6, m = i & 63;
    const int64_t f = f0 + fr;
    float v = 0.f;
    if (f < Tt && f < Tb) v = (db[((int64_t)b * Tt + f) * kMel + m] - mu) * rstd;
    tile[fr][m] = v;
    if (out_btf && f < Tt) Elem<T>::st(out_btf + ((int64_t)b * Tt + f) * kMel + m, v);
  }
  if (out_bft) {
    __syncthreads();
    for (int i = threadIdx.x; i < kFramesPerBlock * kMel; i += 256) {
      const int m = i >> 4, fr = i & 15;
      const int64_t f = f0 + fr;
      if (f < Tt) out_bft[((int64_t)b * kMel + m) * Tt + f] = tile[fr][m];
    }
  }
}

}  // namespace lasr

using namespace lasr;

#ifdef LASR_MEL_STAMPS
extern "C" int lasr_debug_set_mel_stamps(void* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(lasr::g_mel_stamps), &p, sizeof(p));
}
#endif
extern "C" int64_t lasr_mel_num_frames(int64_t n_samples) { return 1 + (n_samples + 2 * kPad) / kHop; }

extern "C" size_t lasr_mel_workspace_bytes(int64_t B, int64_t T) {
  const size_t nblk = (size_t)cdiv(T, kFramesPerBlock);
  return align_up((size_t)B * T * kMel * sizeof(float), 256) + align_up((size_t)B * nblk * 2 * sizeof(double), 256);
}

namespace lasr {

int wave_src_from_c(const lasr_wave_src* s, WaveSrc* out, const char* who) {
  LASR_CHECK_ARG(s && s->wave, "%s: null wave", who);
  LASR_CHECK_ARG(s->wave_dtype == LASR_WAVE_F32 || s->wave_dtype == LASR_WAVE_PCM16, "%s: bad wave dtype %d", who, s->wave_dtype);
  out->wave = s->wave;
  out->pcm16 = s->wave_dtype == LASR_WAVE_PCM16;
  out->dither = s->dither;
  out->dstep = s->dither ? nullptr : reinterpret_cast<const unsigned long long*>(s->dither_step);
  out->dseed = s->dither_seed;
  LASR_CHECK_ARG(s->pitch >= 0, "%s: negative row pitch", who);
  out->pitch = s->pitch;
  return 0;
}

static void launch_norm(int dtype, dim3 grid, hipStream_t st, const float* db, const double* partials, const int32_t* frames_out, int64_t T,
                        int nblk, int normalize, float* out_bft, void* out_btf, const WaveSrc& src) {
  unsigned long long* bump = const_cast<unsigned long long*>(src.dstep);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(mel_norm_kernel<float>, grid, dim3(256), 0, st, db, partials, frames_out, T, nblk, normalize, out_bft,
                       reinterpret_cast<float*>(out_btf), bump);
  else
    hipLaunchKernelGGL(mel_norm_kernel<bf16_t>, grid, dim3(256), 0, st, db, partials, frames_out, T, nblk, normalize, out_bft,
                       reinterpret_cast<bf16_t*>(out_btf), bump);
}

int mel_fwd_src(const WaveSrc& src, const int32_t* sample_lens, const int32_t* aug, int64_t B, int64_t L, int normalize,
                float* out_bft, void* out_btf, int dtype, int32_t* frames_out, float* pct_out, void* workspace,
                size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(src.wave && frames_out && pct_out && workspace, "lasr_mel_fwd: null pointer");
  LASR_CHECK_ARG(out_bft || out_btf, "lasr_mel_fwd: no output requested");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_mel_fwd: bad dtype %d", dtype);
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && L >= 2 && L < (1ll << 30), "lasr_mel_fwd: B=%lld L=%lld", (long long)B, (long long)L);
  const int64_t T = lasr_mel_num_frames(L);
  if (workspace_bytes < lasr_mel_workspace_bytes(B, T)) return fail(LASR_E_WORKSPACE, "lasr_mel_fwd: workspace too small");
  LASR_TRY(init_tables());
  float* db = reinterpret_cast<float*>(workspace);
  double* partials = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + align_up((size_t)B * T * kMel * sizeof(float), 256));
  const int nblk = (int)cdiv(T, kFramesPerBlock);
  dim3 grid(nblk, (unsigned)B);
  hipLaunchKernelGGL(mel_db_kernel, grid, dim3(256), 0, as_stream(stream), src, sample_lens, aug, L, T, db, partials, frames_out, pct_out);
  LASR_LAUNCH_CHECK("mel_db_kernel");
  launch_norm(dtype, grid, as_stream(stream), db, partials, frames_out, T, nblk, normalize, out_bft, out_btf, src);
  LASR_LAUNCH_CHECK("mel_norm_kernel");
  return 0;
}

int ctc_loss_mel_src(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T,
                     int64_t C, int64_t S_max, int blank, float* nll, float* grad, const float* gscale, void* ctc_workspace,
                     size_t ctc_workspace_bytes, const WaveSrc& src, const int32_t* sample_lens, const int32_t* aug, int64_t Bm,
                     int64_t L, int normalize, float* out_bft, void* out_btf, int dtype, int32_t* frames_out, float* pct_out,
                     void* mel_workspace, size_t mel_workspace_bytes, void* stream) {
  LASR_CHECK_ARG(logp && targets && in_lens && tgt_lens && nll && ctc_workspace && src.wave && frames_out && pct_out && mel_workspace,
                 "lasr_ctc_loss_mel: null pointer");
  const int64_t sm = S_max > 0 ? S_max : 1;
  const size_t em_bytes = (size_t)(T + 2) * C * sizeof(float);
  const size_t lds = std::max(sizeof(MelSmem), kCtcMaxS * sizeof(int32_t) + em_bytes);
  static const bool no_fused = getenv("LASR_NO_MEL_CTC") != nullptr;     // A/B switch
  const bool fused = !no_fused && B > 0 && T > 0 && C > 1 && 2 * S_max + 1 <= 256 && lds <= 80 * 1024 && C % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(logp) % 16 == 0 && !getenv("LASR_CTC_NO_LDS") && (out_bft || out_btf) &&
                     (dtype == LASR_F32 || dtype == LASR_BF16) && Bm > 0 && Bm < 65536 && L >= 2 && L < (1ll << 30) &&
                     ctc_workspace_bytes >= lasr_ctc_workspace_bytes(B, T, S_max) &&
                     mel_workspace_bytes >= lasr_mel_workspace_bytes(Bm, lasr_mel_num_frames(L));
  if (!fused) {
    LASR_TRY(lasr_ctc_loss(logp, targets, in_lens, tgt_lens, B, T, C, S_max, blank, nll, grad, gscale, ctc_workspace, ctc_workspace_bytes, stream));
    return mel_fwd_src(src, sample_lens, aug, Bm, L, normalize, out_bft, out_btf, dtype, frames_out, pct_out, mel_workspace,
                       mel_workspace_bytes, stream);
  }
  LASR_CHECK_SHAPE(blank >= 0 && blank < C, "lasr_ctc_loss_mel: blank");
  LASR_TRY(init_tables());
  const int64_t Tm = lasr_mel_num_frames(L);
  MelCtcArgs a;
  const size_t ab = (size_t)B * T * 64 * 4;
  a.logp = logp; a.targets = targets; a.in_lens = in_lens; a.tgt_lens = tgt_lens; a.T = T; a.C = C; a.S_max = sm; a.blank = blank;
  a.alpha = reinterpret_cast<float*>(ctc_workspace);
  a.beta = a.alpha + ab;
  a.next_same = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ctc_workspace) + align_up(2 * ab * sizeof(float), 256));
  a.nll = nll; a.n_ctc = (int)B;
  a.src = src; a.sample_lens = sample_lens; a.aug = aug; a.L = L; a.Tm = Tm;
  a.db_out = reinterpret_cast<float*>(mel_workspace);
  a.partials = reinterpret_cast<double*>(reinterpret_cast<char*>(mel_workspace) + align_up((size_t)Bm * Tm * kMel * sizeof(float), 256));
  a.frames_out = frames_out; a.pct_out = pct_out;
  const int nblk = (int)cdiv(Tm, kFramesPerBlock);
  a.nbx = nblk;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mel_ctc_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipLaunchKernelGGL(mel_ctc_kernel<4>, dim3((unsigned)(B + (int64_t)nblk * Bm)), dim3(256), lds, as_stream(stream), a);
  LASR_LAUNCH_CHECK("mel_ctc_kernel");
  if (grad) LASR_TRY(launch_ctc_grad(logp, targets, in_lens, tgt_lens, B, T, C, S_max, blank, nll, grad, gscale, ctc_workspace, stream));
  dim3 grid(nblk, (unsigned)Bm);
  launch_norm(dtype, grid, as_stream(stream), a.db_out, a.partials, frames_out, Tm, nblk, normalize, out_bft, out_btf, src);
  LASR_LAUNCH_CHECK("mel_norm_kernel");
  return 0;
}

bool compact_lattice_mel_fits(int64_t T, int64_t CE) {
  static const bool off = getenv("LASR_NO_MEL_CTC") != nullptr || getenv("LASR_CTC_NO_LDS") != nullptr;
  return !off && kCtcMaxS * sizeof(int32_t) + (size_t)(T + 2) * CE * sizeof(float) <= 158 * 1024;
}

int launch_compact_lattice_mel(const float* E, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T,
                               int64_t CE, int64_t S_max, int blank_col, float* alpha, float* beta, int32_t* next_same, float* nll, int ns,
                               const MelJob& job, void* stream) {
  LASR_CHECK_ARG(E && job.src.wave && job.out_btf && job.frames_out && job.pct_out && job.ws, "lasr_ctc_loss_lean (+features): null pointer");
  LASR_CHECK_ARG(job.dtype == LASR_F32 || job.dtype == LASR_BF16, "lasr_ctc_loss_lean (+features): bad dtype");
  LASR_CHECK_SHAPE(job.B > 0 && job.B < 65536 && job.L >= 2 && job.L < (1ll << 30), "lasr_ctc_loss_lean (+features): B=%lld L=%lld", (long long)job.B,
                   (long long)job.L);
  const int64_t Tm = lasr_mel_num_frames(job.L);
  if (job.ws_bytes < lasr_mel_workspace_bytes(job.B, Tm)) return fail(LASR_E_WORKSPACE, "lasr_ctc_loss_lean (+features): workspace");
  LASR_TRY(init_tables());
  MelCtcArgs a;
  a.logp = E; a.targets = targets; a.in_lens = in_lens; a.tgt_lens = tgt_lens; a.T = T; a.C = CE; a.S_max = S_max; a.blank = blank_col;
  a.alpha = alpha; a.beta = beta; a.next_same = next_same; a.nll = nll; a.n_ctc = (int)B;
  a.src = job.src; a.sample_lens = job.sample_lens; a.aug = job.aug; a.L = job.L; a.Tm = Tm;
  a.db_out = reinterpret_cast<float*>(job.ws);
  a.partials = reinterpret_cast<double*>(reinterpret_cast<char*>(job.ws) + align_up((size_t)job.B * Tm * kMel * sizeof(float), 256));
  a.frames_out = job.frames_out; a.pct_out = job.pct_out;
  const int nblk = (int)cdiv(Tm, kFramesPerBlock);
  a.nbx = nblk;
  const size_t lds = std::max(sizeof(MelSmem), kCtcMaxS * sizeof(int32_t) + (size_t)(T + 2) * CE * sizeof(float));
  const dim3 grid((unsigned)(B + (int64_t)nblk * job.B));
#define LASR_MCK(NS_)                                                                                                              \
  do {                                                                                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mel_ctc_kernel<NS_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((mel_ctc_kernel<NS_, true>), grid, dim3(256), lds, as_stream(stream), a);                                     \
  } while (0)
  if (ns == 4) LASR_MCK(4); else if (ns == 8) LASR_MCK(8); else LASR_MCK(16);
#undef LASR_MCK
  LASR_LAUNCH_CHECK("mel_ctc_kernel (compact)");
  launch_norm(job.dtype, dim3(nblk, (unsigned)job.B), as_stream(stream), a.db_out, a.partials, job.frames_out, Tm, nblk, job.normalize, nullptr,
              job.out_btf, job.src);
  LASR_LAUNCH_CHECK("mel_norm_kernel");
  return 0;
}

__global__ __launch_bounds__(256) void dither_noise_kernel(unsigned long long seed, const unsigned long long* __restrict__ step,
                                                           int64_t L, float* __restrict__ out) {
  const uint32_t b = blockIdx.y;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (4 * g >= L) return;
  float z[4];
  dither4(seed, *step, b, (uint32_t)g, z);
  for (int q = 0; q < 4; ++q)
    if (4 * g + q < L) out[(int64_t)b * L + 4 * g + q] = z[q];
}

// SpecAugment's zeros on an existing (B, F, T) f32 feature tensor (data_module.py:97-122): rows [rx, rx+wx) and frames [ry, ry+wy)
__global__ __launch_bounds__(256) void spec_augment_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const int32_t* __restrict__ aug, int64_t F, int64_t T) {
  const int64_t b = blockIdx.z, f = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int rx = aug[b * 4 + 0], wx = aug[b * 4 + 1], ry = aug[b * 4 + 2], wy = aug[b * 4 + 3];
  const int64_t i = (b * F + f) * T + t;
  const bool hit = (f >= rx && f < rx + wx) || (t >= ry && t < ry + wy);
  out[i] = hit ? 0.f : in[i];
}

}  // namespace lasr

extern "C" int lasr_spec_augment(const float* in, float* out, const int32_t* aug, int64_t B, int64_t F, int64_t T, void* stream) {
  LASR_CHECK_ARG(in && out && aug, "lasr_spec_augment: null pointer");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && F > 0 && F < 65536 && T > 0 && T < (1ll << 30), "lasr_spec_augment: B=%lld F=%lld T=%lld",
                   (long long)B, (long long)F, (long long)T);
  hipLaunchKernelGGL(spec_augment_kernel, dim3((unsigned)cdiv(T, 256), (unsigned)F, (unsigned)B), dim3(256), 0, as_stream(stream),
                     in, out, aug, F, T);
  LASR_LAUNCH_CHECK("spec_augment_kernel");
  return 0;
}

extern "C" int lasr_mel_fwd(const float* wave, const int32_t* sample_lens, const float* dither, const int32_t* aug,
                            int64_t B, int64_t L, int normalize, float* out_bft, void* out_btf, int dtype,
                            int32_t* frames_out, float* pct_out, void* workspace, size_t workspace_bytes, void* stream) {
  const WaveSrc src = {wave, 0, dither, nullptr, 0ull, 0};
  return mel_fwd_src(src, sample_lens, aug, B, L, normalize, out_bft, out_btf, dtype, frames_out, pct_out, workspace, workspace_bytes, stream);
}

extern "C" int lasr_mel_fwd_src(const lasr_wave_src* wsrc, const int32_t* sample_lens, const int32_t* aug, int64_t B, int64_t L,
                                int normalize, float* out_bft, void* out_btf, int dtype, int32_t* frames_out, float* pct_out,
                                void* workspace, size_t workspace_bytes, void* stream) {
  WaveSrc src;
  LASR_TRY(wave_src_from_c(wsrc, &src, "lasr_mel_fwd_src"));
  return mel_fwd_src(src, sample_lens, aug, B, L, normalize, out_bft, out_btf, dtype, frames_out, pct_out, workspace, workspace_bytes, stream);
}

extern "C" int lasr_dither_noise(uint64_t seed, const uint64_t* step, int64_t B, int64_t L, float* out, void* stream) {
  LASR_CHECK_ARG(step && out, "lasr_dither_noise: null pointer");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && L >= 1 && L < (1ll << 30), "lasr_dither_noise: B=%lld L=%lld", (long long)B, (long long)L);
  hipLaunchKernelGGL(dither_noise_kernel, dim3((unsigned)cdiv(cdiv(L, 4), 256), (unsigned)B), dim3(256), 0, as_stream(stream),
                     (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(step), L, out);
  LASR_LAUNCH_CHECK("dither_noise_kernel");
  return 0;
}

extern "C" int lasr_ctc_loss_mel(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B,
                                 int64_t T, int64_t C, int64_t S_max, int blank, float* nll, float* grad, const float* gscale,
                                 void* ctc_workspace, size_t ctc_workspace_bytes, const float* wave, const int32_t* sample_lens,
                                 const float* dither, const int32_t* aug, int64_t Bm, int64_t L, int normalize, float* out_bft,
                                 void* out_btf, int dtype, int32_t* frames_out, float* pct_out, void* mel_workspace,
                                 size_t mel_workspace_bytes, void* stream) {
  const WaveSrc src = {wave, 0, dither, nullptr, 0ull, 0};
  return ctc_loss_mel_src(logp, targets, in_lens, tgt_lens, B, T, C, S_max, blank, nll, grad, gscale, ctc_workspace, ctc_workspace_bytes, src,
                          sample_lens, aug, Bm, L, normalize, out_bft, out_btf, dtype, frames_out, pct_out, mel_workspace,
                          mel_workspace_bytes, stream);
}
