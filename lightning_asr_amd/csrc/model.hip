// Native execution plan for MyModel2 (models/QuartNet.py:264-291 and the Context / ContextSE
// variants): the library sequences every kernel of forward, CTC loss and backward on one stream
// over caller-owned flat buffers.  No tracing, no autograd: the plan is static per (B, T_in), so
// the whole step can be captured into a hipGraph by the host.
#include "common.h"
#include "mel.h"
#include "fused.h"
#include "gemm.h"
#include "lstm_body.h"
#include "se_seqsum.h"
#include <string>
#include <vector>
#include <math.h>

extern "C" size_t lasr_bn_bwd_workspace_bytes(int64_t B, int64_t T, int64_t C);
extern "C" size_t lasr_gemm_workspace_bytes(int64_t M, int64_t N, int split_k, int want_stats);
extern "C" size_t lasr_dwconv_wgrad_workspace_bytes(int64_t B, int64_t Tout, int64_t C, int k);
extern "C" int lasr_log_softmax_bwd(const float*, const float*, float*, int64_t, int64_t, void*);
extern "C" int lasr_cast_f32_to_bf16(const float*, void*, int64_t, void*);
extern "C" int lasr_colsum_f32(const float* x, float* out, int64_t rows, int64_t C, void* workspace, size_t workspace_bytes, void* stream);
extern "C" size_t lasr_colsum_workspace_bytes(int64_t rows, int64_t C);
extern "C" int lasr_scale_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream);
extern "C" int lasr_seqsum(const void* x, int dtype, int64_t B, int64_t T, int64_t C, float* sums, void* stream);
extern "C" int lasr_se_fwd(const float* sums, const float* coef, const float* W1, const float* W2, int64_t B, int64_t T, int64_t C,
                           float* pooled, float* hidden, float* scale, void* stream);
extern "C" size_t lasr_se_bwd_workspace_bytes(int64_t B, int64_t C);
extern "C" int lasr_se_bwd(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                           const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T,
                           int64_t C, int act, float* seg, float* dW1, float* dW2, void* workspace, size_t workspace_bytes, void* stream);
extern "C" size_t lasr_bilstm_saved_bytes(int64_t B, int64_t T);
extern "C" int lasr_bilstm_fwd(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const float* bih_f,
                               const float* bhh_f, const float* bih_r, const float* bhh_r, const int32_t* lens, int64_t B, int64_t T,
                               void* out, int dtype, int64_t ld_out, int64_t col0, float* saved, void* stream);
extern "C" size_t lasr_bilstm_bwd_workspace_bytes(int64_t B);
extern "C" int lasr_bilstm_bwd(const void* dout, int dtype, int64_t ld_dout, int64_t col0, const float* whh_f, const float* whh_r,
                               const int32_t* lens, int64_t B, int64_t T, const float* saved, float* dg_f, float* dg_r, float* dwhh_f,
                               float* dwhh_r, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int lasr_se_bwd_drop(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                                const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T,
                                int64_t C, int act, const lasr_dropout* dropout, float* seg, float* dW1, float* dW2, void* workspace,
                                size_t workspace_bytes, void* stream);
extern "C" size_t lasr_bn_se_bwd_workspace_bytes(int64_t B, int64_t T, int64_t C);
extern "C" int lasr_bn_se_bwd(const void* dout, const void* y, const float* coef, const float* saved, const float* gamma, const float* beta,
                              const void* y2, const float* coef2, const float* saved2, const float* gamma2, const float* se_scale,
                              const float* se_hidden, const float* se_pooled, const float* ysum, const float* W1, const float* W2,
                              const int32_t* row_lens, void* dy, void* dy2, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2,
                              float* dW1, float* dW2, float* seg_out, int dtype, int64_t B, int64_t T, int64_t C, int act,
                              const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int lasr_gemm_rowstat(const void* A, const void* B, const float* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                 float* row_stat, int32_t* row_arg, int* n_col_tiles, void* stream);
extern "C" size_t lasr_gemm_rowstat_bytes(int64_t M, int64_t N);
extern "C" size_t lasr_ctc_lean_workspace_bytes(int64_t B, int64_t T, int64_t C, int64_t S_max);
extern "C" int lasr_ctc_loss_lean(const void* logits, int64_t ldc, const float* row_stat, const int32_t* row_arg, int n_col_tiles,
                                  const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T, int64_t C,
                                  int64_t S_max, int blank, float* nll, int32_t* argmax, void* grad, float* bias_grad, const float* gscale,
                                  void* workspace, size_t workspace_bytes, void* stream);
extern "C" int lasr_copy_cols(const void* src, int src_dtype, int64_t ld_src, int64_t scol0, void* dst, int dst_dtype, int64_t ld_dst,
                              int64_t dcol0, int64_t rows, int64_t ncols, int accumulate, void* stream);

namespace lasr {

static constexpr float kBnEps = 1e-3f;   // nn.BatchNorm1d(out_ch, eps=1e-3)  models/QuartNet.py:24
static constexpr float kBnMom = 0.1f;

struct TensorInfo {
  std::string name;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0;
  int kind = 0;  // 0 param, 1 f32 buffer, 2 int64 counter
  int64_t offset = 0;
  int64_t numel = 1;
};

struct BnRef { int64_t gamma = -1, beta = -1, rmean = -1, rvar = -1; };

// split-K of the 1x1 weight gradients (K = B*T' rows): 16 slices x 16 tiles x 2 problems = 512 workgroups
static int wgrad_split() { static const int v = getenv("LASR_WGRAD_SPLIT") ? atoi(getenv("LASR_WGRAD_SPLIT")) : 16; return v < 1 ? 1 : (v > 64 ? 64 : v); }
static int dec_split_k() { static const int v = getenv("LASR_DEC_SPLIT") ? atoi(getenv("LASR_DEC_SPLIT")) : 4; return v < 1 ? 1 : (v > 16 ? 16 : v); }
static bool no_fuse() { static const bool v = getenv("LASR_NO_FUSE") != nullptr; return v; }

// One "unit": [depthwise conv] -> 1x1 GEMM (+mask) -> BN  [+ residual 1x1 GEMM -> BN] -> activation
struct Unit {
  std::string tap;
  std::string rx_fwd, rx_bwd;     // roctx range names "lasr:fwd <tap>" / "lasr:bwd <tap>" (made once, at model creation)
  int ci = 0, co = 0, k = 0, stride = 1;
  bool has_dw = false, has_res = false, masked = false, act = true, has_se = false, ctx_before = false;
  int64_t w_dw = -1, w_pw = -1, w_res = -1, w_se1 = -1, w_se2 = -1;
  BnRef bn, bn_res;
  // workspace offsets (bytes) filled by plan()
  size_t o_u = 0, o_y = 0, o_y2 = 0, o_out = 0, o_coef = 0, o_saved = 0, o_coef2 = 0, o_saved2 = 0, o_stats = 0, o_stats2 = 0;
  size_t o_se_sum = 0, o_se_pool = 0, o_se_hid = 0, o_se_scale = 0, o_se_grad = 0;   // SE: [B][co] (hid: [B][co/8]) f32
  size_t o_wfold = 0, o_bfold = 0;   // eval: folded [a W | a2 Wr] bf16 and b + b2 (residual units)
  size_t o_wgp = 0, wgp_bytes = 0, o_dwp = 0, dwp_bytes = 0, o_dy = 0, o_dy2 = 0;   // deferred reductions: split-K slabs of dW / dWr, depthwise-dW partials
  size_t o_taps = 0;     // bf16, stride-1 depthwise layers of 64 k channels: the step's tap tables [2][ci][kDwTapRow] (fused.h); 0 = none
};

// BiLSTM context branch (Context / ContextSE): parameter offsets per direction
struct LstmRef { int64_t w_ih[2] = {-1, -1}, w_hh[2] = {-1, -1}, b_ih[2] = {-1, -1}, b_hh[2] = {-1, -1}; };

struct Plan {
  int64_t B = 0, T_in = 0, T = 0, S_max = 0;
  size_t total = 0;
  size_t o_lens = 0, o_logits = 0, o_glogits = 0, o_nll = 0, o_scratch = 0, o_g[2] = {0, 0}, o_d1 = 0, o_d2 = 0, o_du = 0, o_dxr = 0;
  size_t o_sums = 0, o_sums2 = 0, o_ctc = 0, o_wbf16 = 0;
  size_t o_cat = 0, o_gx[2] = {0, 0}, o_lstm_saved = 0, o_dg[2] = {0, 0};   // context: [N][336] | [N][160] f32 x2 | saved | [N][160] f32 x2
  size_t o_dgb[2] = {0, 0}, o_lstm_part = 0, o_lstm_bpart = 0, o_lstm_wgp = 0, lstm_wgp_bytes = 0;   // bf16 mode: dg as bf16 x2 | dW_hh partials | bias partials | dW_ih slabs
  size_t scratch_bytes = 0, ctc_bytes = 0;
  size_t o_rowstat = 0, o_lean = 0, lean_bytes = 0;   // large-vocabulary head: softmax row statistics, lasr_ctc_loss_lean workspace
};

}  // namespace lasr

using namespace lasr;

struct lasr_model {
  lasr_model_config cfg;
  std::vector<TensorInfo> tensors;
  std::vector<Unit> units;
  int64_t n_param = 0, n_buffer = 0;
  int64_t w_dec = -1, b_dec = -1;
  LstmRef lstm;
  Plan plan;
  bool planned = false;
  // nn.Dropout(p = drop_rate) of every SeprationConv and of last_cnn2 (models/QuartNet.py:26,38,149): counter-based masks
  // regenerated in forward and backward (dropout.h); drop_step is a device scalar the first launch of a training forward bumps
  float drop_p = 0.f;
  uint64_t drop_seed = 0;
  uint64_t* drop_step = nullptr;
  const float* tail_nll = nullptr; float* tail_loss = nullptr;   // the batch mean of the losses, left by forward_and_loss to the backward's first launch
  bool lean_active = false;   // the last loss ran the large-vocabulary head: d(logits) is the bf16 [N][ldc] tensor at o_d1, db is done
  int lean_tiles = 0;
  int bwd_cur = 0;       // ping-pong index of the gradient buffers between partial backward calls
  bool taps_valid = false;   // the workspace holds this step's depthwise tap tables (made by the last training forward)
  int bwd_next = -1;     // next unit a lasr_model_backward_continue call would process (-1: nothing pending)
  // one-shot feature prefetch consumed by the next loss_backward call (lasr_model_set_prefetch)
  struct Prefetch {
    bool armed = false;
    WaveSrc src; const int32_t* sample_lens; const int32_t* aug;
    int64_t B, L; int normalize; void* out_btf; int dtype; int32_t* frames_out; float* pct_out; void* ws; size_t ws_bytes;
  } prefetch;

  int64_t add_tensor(const std::string& name, std::initializer_list<int64_t> shape, int kind) {
    TensorInfo t;
    t.name = name;
    t.kind = kind;
    t.ndim = (int)shape.size();
    int i = 0;
    for (int64_t s : shape) { t.shape[i++] = s; t.numel *= s; }
    if (kind == 0) { t.offset = n_param; n_param += t.numel; }
    else if (kind == 1) { t.offset = n_buffer; n_buffer += t.numel; }
    tensors.push_back(t);
    return t.offset;
  }
  BnRef add_bn(const std::string& p, int64_t c) {
    BnRef r;
    r.gamma = add_tensor(p + ".weight", {c}, 0);
    r.beta = add_tensor(p + ".bias", {c}, 0);
    r.rmean = add_tensor(p + ".running_mean", {c}, 1);
    r.rvar = add_tensor(p + ".running_var", {c}, 1);
    add_tensor(p + ".num_batches_tracked", {}, 2);
    return r;
  }
  void add_sep(Unit& u, const std::string& p) {
    u.w_dw = add_tensor(p + ".depthwise_conv.weight", {u.ci, 1, u.k}, 0);
    u.w_pw = add_tensor(p + ".pointwise_conv.weight", {u.co, u.ci, 1}, 0);
    u.bn = add_bn(p + ".bn", u.co);
    if (cfg.variant == LASR_VARIANT_CONTEXT_SE) {   // SELayer(out_ch, reduction=8)  models/QuartNetContextSE.py:46
      u.has_se = true;
      u.w_se1 = add_tensor(p + ".se.fc.0.weight", {u.co / 8, u.co}, 0);
      u.w_se2 = add_tensor(p + ".se.fc.2.weight", {u.co, u.co / 8}, 0);
    }
  }
};

static int build_model(lasr_model* m) {
  const lasr_model_config& c = m->cfg;
  const bool ctx = c.variant != LASR_VARIANT_PLAIN;
  {  // first_cnn: SeprationConv(in_c, 256, k=33, stride=2)            models/QuartNet.py:129
    Unit u;
    u.tap = "first_cnn"; u.ci = c.in_c; u.co = 256; u.k = 33; u.stride = 2; u.has_dw = true; u.masked = c.mask != 0;
    m->add_sep(u, "encoder.first_cnn");
    m->units.push_back(u);
  }
  struct B { const char* name; int ci, co, k; };
  std::vector<B> blocks = {{"block1", 256, 256, 33}, {"block12", 256, 256, 33}, {"block13", 256, 256, 33},
                           {"block2", 256, 256, 39}, {"block22", 256, 256, 39}, {"block23", 256, 256, 39},
                           {"block3", ctx ? 336 : 256, 512, 51}, {"block32", 512, 512, 51}, {"block33", 512, 512, 51},
                           {"block4", 512, 512, 63}, {"block42", 512, 512, 63}, {"block43", 512, 512, 63},
                           {"block5", 512, 512, 75}};
  if (ctx) blocks.push_back({"block6", 512, 512, 87});
  for (const B& b : blocks) {  // QuartNetBlock(repeat=1)                 models/QuartNet.py:55-78
    Unit u;
    u.tap = b.name; u.ci = b.ci; u.co = b.co; u.k = b.k; u.has_dw = true; u.has_res = true; u.masked = c.mask != 0;
    u.ctx_before = ctx && std::string(b.name) == "block3";   // cat(x, BiLSTM(x)) feeds block3  (QuartNetContext.py:171-174)
    const std::string p = std::string("encoder.") + b.name;
    u.w_res = m->add_tensor(p + ".reside.0.weight", {u.co, u.ci, 1}, 0);
    u.bn_res = m->add_bn(p + ".reside.1", u.co);
    m->add_sep(u, p + ".seq.0");
    m->units.push_back(u);
  }
  {  // last_cnn2: 1x1 512->1024 (no bias) + BN + ReLU, never masked     models/QuartNet.py:145-150
    Unit u;
    u.tap = "last_cnn2"; u.ci = 512; u.co = 1024;
    u.w_pw = m->add_tensor("encoder.last_cnn2.0.weight", {1024, 512, 1}, 0);
    u.bn = m->add_bn("encoder.last_cnn2.1", 1024);
    m->units.push_back(u);
  }
  for (Unit& u : m->units) { u.rx_fwd = "lasr:fwd " + u.tap; u.rx_bwd = "lasr:bwd " + u.tap; }
  if (ctx) {  // context_rnn = nn.LSTM(256, 40, bidirectional)   models/QuartNetContext.py:157
    const std::string r = "encoder.context_rnn.rnn.";
    for (int d = 0; d < 2; ++d) {
      const std::string sfx = d ? "_reverse" : "";
      m->lstm.w_ih[d] = m->add_tensor(r + "weight_ih_l0" + sfx, {160, 256}, 0);
      m->lstm.w_hh[d] = m->add_tensor(r + "weight_hh_l0" + sfx, {160, 40}, 0);
      m->lstm.b_ih[d] = m->add_tensor(r + "bias_ih_l0" + sfx, {160}, 0);
      m->lstm.b_hh[d] = m->add_tensor(r + "bias_hh_l0" + sfx, {160}, 0);
    }
  }
  m->w_dec = m->add_tensor("decoder.weight", {c.n_class, 1024, 1}, 0);
  m->b_dec = m->add_tensor("decoder.bias", {c.n_class}, 0);
  return 0;
}

// The large-vocabulary head (bf16 logits + row statistics instead of f32 logits / log-probs / gradients) pays when the class
// axis dwarfs the label axis: C >= 256 (AISHELL: 4334).  LASR_NO_LEAN_HEAD=1 keeps the dense head for A/B runs.
static bool lean_capable(const lasr_model* m) {
  static const bool off = getenv("LASR_NO_LEAN_HEAD") != nullptr;
  return !off && m->cfg.dtype == LASR_BF16 && m->cfg.n_class >= 256 && m->cfg.n_class <= 9216;
}
static int64_t lean_ldc(const lasr_model* m) { return ((int64_t)m->cfg.n_class + 7) / 8 * 8; }

static int64_t out_frames(int64_t T_in) { return (T_in + 2 * 16 - 33) / 2 + 1; }

static size_t take(size_t& cur, size_t bytes) {
  const size_t o = cur;
  cur += align_up(bytes, 256);
  return o;
}

static void make_plan(lasr_model* m, int64_t B, int64_t T_in, int64_t S_max) {
  Plan& p = m->plan;
  if (m->planned && p.B == B && p.T_in == T_in && p.S_max == S_max) return;
  p = Plan();
  p.B = B; p.T_in = T_in; p.T = out_frames(T_in); p.S_max = S_max;
  const size_t es = dtype_size(m->cfg.dtype);
  const int64_t N = B * p.T;
  size_t cur = 0;
  p.o_lens = take(cur, (size_t)B * sizeof(int32_t));
  size_t scratch = 0;
  int64_t cmax = 0;
  for (Unit& u : m->units) {
    if (u.has_dw) u.o_u = take(cur, (size_t)N * u.ci * es);
    u.o_y = take(cur, (size_t)N * u.co * es);
    if (u.has_res) u.o_y2 = take(cur, (size_t)N * u.co * es);
    u.o_out = take(cur, (size_t)N * u.co * es);
    u.o_coef = take(cur, 2 * u.co * sizeof(float));
    u.o_saved = take(cur, 2 * u.co * sizeof(float));
    u.o_stats = take(cur, 2 * u.co * sizeof(float));
    if (u.has_res) {
      u.o_coef2 = take(cur, 2 * u.co * sizeof(float));
      u.o_saved2 = take(cur, 2 * u.co * sizeof(float));
      u.o_stats2 = take(cur, 2 * u.co * sizeof(float));
    }
    if (u.has_se) {
      u.o_se_sum = take(cur, (size_t)B * u.co * sizeof(float));
      u.o_se_pool = take(cur, (size_t)B * u.co * sizeof(float));
      u.o_se_hid = take(cur, (size_t)B * (u.co / 8) * sizeof(float));
      u.o_se_scale = take(cur, (size_t)B * u.co * sizeof(float));
      u.o_se_grad = take(cur, (size_t)B * u.co * sizeof(float));
      scratch = std::max(scratch, lasr_se_bwd_workspace_bytes(B, u.co));
      scratch = std::max(scratch, lasr_bn_se_bwd_workspace_bytes(B, p.T, u.co));
    }
    cmax = std::max<int64_t>(cmax, std::max(u.ci, u.co));
    scratch = std::max(scratch, 2 * lasr_gemm_workspace_bytes(N, u.co, 1, 1));
    scratch = std::max(scratch, lasr_bn_bwd_workspace_bytes(B, p.T, u.co));
    scratch = std::max(scratch, 2 * lasr_gemm_workspace_bytes(u.co, u.ci, wgrad_split(), 0));
    if (u.has_dw) scratch = std::max(scratch, lasr_dwconv_wgrad_workspace_bytes(B, p.T, u.ci, u.k));
  }
  const int64_t C = m->cfg.n_class;
  scratch = std::max(scratch, lasr_gemm_workspace_bytes(C, 1024, 16, 0));
  scratch = std::max(scratch, lasr_gemm_workspace_bytes(N, C, 16, 0));   // decoder forward, split-K partials
  scratch = std::max(scratch, lasr_colsum_workspace_bytes(N, C));
  if (m->cfg.variant != LASR_VARIANT_PLAIN) {
    p.o_cat = take(cur, (size_t)N * 336 * es);
    for (int d = 0; d < 2; ++d) { p.o_gx[d] = take(cur, (size_t)N * 160 * sizeof(float)); p.o_dg[d] = take(cur, (size_t)N * 160 * sizeof(float)); }
    p.o_lstm_saved = take(cur, lasr_bilstm_saved_bytes(B, p.T));
    scratch = std::max(scratch, lasr_bilstm_bwd_workspace_bytes(B));
    scratch = std::max(scratch, lasr_gemm_workspace_bytes(160, 256, 16, 0));
    scratch = std::max(scratch, lasr_colsum_workspace_bytes(N, 160));
    if (m->cfg.dtype == LASR_BF16) {   // what the recurrences' launch leaves for the stage's closing launch / reduction (backward_from_glogits)
      for (int d = 0; d < 2; ++d) p.o_dgb[d] = take(cur, (size_t)N * 160 * sizeof(bf16_t));
      p.o_lstm_part = take(cur, (size_t)2 * B * lstm::kDwZ * lstm::G * lstm::H * sizeof(float));
      p.o_lstm_bpart = take(cur, (size_t)2 * B * lstm::kDwZ * lstm::G * sizeof(float));
      p.lstm_wgp_bytes = lasr_gemm_workspace_bytes(160, 256, wgrad_split(), 0);
      p.o_lstm_wgp = take(cur, 2 * p.lstm_wgp_bytes);
    }
  }
  // f32 logits (dense head) or bf16 logits with rows padded to 8 classes (large-vocabulary head) - and, in the SAME bytes, the dense
  // head's d(loss)/d(logits): log_softmax has consumed the logits before the CTC gradient kernel writes, and backward reads only the
  // gradient (444 MB less at C = 4334, bs = 32; tap "logits" of a dense TRAINING step therefore shows the gradient)
  p.o_logits = take(cur, (size_t)N * C * sizeof(float));
  p.o_glogits = p.o_logits;
  p.o_nll = take(cur, (size_t)(B + 1) * sizeof(float));
  for (Unit& u : m->units) {   // eval-mode folded weights
    if (u.has_res) {
      u.o_wfold = take(cur, (size_t)u.co * 2 * u.ci * sizeof(bf16_t));
      u.o_bfold = take(cur, (size_t)u.co * sizeof(float));
    }
  }
  for (Unit& u : m->units) {   // slabs that outlive `scratch`: summed by one lasr_reduce_many per backward stage
    {
      u.wgp_bytes = (u.has_res ? 2 : 1) * lasr_gemm_workspace_bytes(u.co, u.ci, wgrad_split(), 0);
      u.o_wgp = take(cur, u.wgp_bytes);
      // the unit's dy / dy2 outlive its backward: their weight-gradient GEMMs run batched at the end of the stage
      u.o_dy = take(cur, (size_t)N * u.co * es);
      if (u.has_res) u.o_dy2 = take(cur, (size_t)N * u.co * es);
    }
    if (u.has_dw) {
      u.dwp_bytes = lasr_dwconv_wgrad_workspace_bytes(B, p.T, u.ci, u.k);
      u.o_dwp = take(cur, u.dwp_bytes);
      u.o_taps = 0;
      if (m->cfg.dtype == LASR_BF16 && u.stride == 1 && u.ci % 64 == 0 && dw_taps_enabled())
        u.o_taps = take(cur, (size_t)2 * u.ci * kDwTapRow * sizeof(uint32_t));
    }
  }
  p.scratch_bytes = scratch;
  p.o_scratch = take(cur, scratch);
  p.o_g[0] = take(cur, (size_t)N * cmax * es);
  p.o_g[1] = take(cur, (size_t)N * cmax * es);
  p.o_d1 = take(cur, (size_t)N * std::max<int64_t>(cmax, (C + 7) / 8 * 8) * es);   // dy, or the row-padded bf16 logit gradient
  p.o_d2 = take(cur, (size_t)N * cmax * es);
  p.o_du = take(cur, (size_t)N * cmax * es);
  p.o_dxr = take(cur, (size_t)N * cmax * es);
  p.o_sums = take(cur, 2 * cmax * sizeof(float));
  p.o_sums2 = take(cur, 2 * cmax * sizeof(float));
  p.ctc_bytes = lasr_ctc_workspace_bytes(B, p.T, S_max);
  p.o_ctc = take(cur, p.ctc_bytes);
  if (lean_capable(m)) {
    p.o_rowstat = take(cur, lasr_gemm_rowstat_bytes(N, C));
    p.lean_bytes = lasr_ctc_lean_workspace_bytes(B, p.T, C, S_max);
    p.o_lean = take(cur, p.lean_bytes);
  }
  if (m->cfg.dtype == LASR_BF16) p.o_wbf16 = take(cur, (size_t)m->n_param * sizeof(bf16_t));
  p.total = cur;
  m->planned = true;
}

extern "C" int lasr_model_create(const lasr_model_config* cfg, lasr_model_t** out) {
  LASR_CHECK_ARG(cfg && out, "lasr_model_create: null pointer");
  LASR_CHECK_ARG(cfg->variant >= 0 && cfg->variant <= 2 && cfg->n_class >= 2 && cfg->in_c > 0 && cfg->in_c % 4 == 0,
                 "lasr_model_create: bad config (variant=%d n_class=%d in_c=%d)", cfg->variant, cfg->n_class, cfg->in_c);
  LASR_CHECK_ARG(cfg->dtype == LASR_F32 || cfg->dtype == LASR_BF16, "lasr_model_create: bad dtype");
  LASR_CHECK_ARG(cfg->act == LASR_ACT_RELU || cfg->act == LASR_ACT_SWISH, "lasr_model_create: bad activation");
  lasr_model* m = new lasr_model();
  m->cfg = *cfg;
  int rc = build_model(m);
  if (rc) { delete m; return rc; }
  *out = m;
  return 0;
}

extern "C" void lasr_model_destroy(lasr_model_t* m) { delete m; }

extern "C" int lasr_model_set_dropout(lasr_model_t* m, float p, uint64_t seed, uint64_t* step_counter) {
  LASR_CHECK_ARG(m && p >= 0.f && p < 1.f && (p == 0.f || step_counter), "lasr_model_set_dropout: 0 <= p < 1 and a device step counter");
  m->drop_p = p; m->drop_seed = seed; m->drop_step = p > 0.f ? step_counter : nullptr;
  return 0;
}

static int set_prefetch(lasr_model_t* m, const WaveSrc& src, const int32_t* sample_lens, const int32_t* aug, int64_t B, int64_t L,
                        int normalize, void* out_btf, int dtype, int32_t* frames_out, float* pct_out, void* mel_workspace,
                        size_t mel_workspace_bytes) {
  LASR_CHECK_ARG(m && src.wave && out_btf && frames_out && pct_out && mel_workspace, "lasr_model_set_prefetch: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_model_set_prefetch: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && L >= 2 && L < (1ll << 30), "lasr_model_set_prefetch: B=%lld L=%lld", (long long)B, (long long)L);
  if (mel_workspace_bytes < lasr_mel_workspace_bytes(B, lasr_mel_num_frames(L))) return fail(LASR_E_WORKSPACE, "lasr_model_set_prefetch: workspace");
  m->prefetch.armed = true;
  m->prefetch.src = src; m->prefetch.sample_lens = sample_lens; m->prefetch.aug = aug;
  m->prefetch.B = B; m->prefetch.L = L; m->prefetch.normalize = normalize; m->prefetch.out_btf = out_btf; m->prefetch.dtype = dtype;
  m->prefetch.frames_out = frames_out; m->prefetch.pct_out = pct_out; m->prefetch.ws = mel_workspace; m->prefetch.ws_bytes = mel_workspace_bytes;
  return 0;
}

extern "C" int lasr_model_set_prefetch(lasr_model_t* m, const float* wave, const int32_t* sample_lens, const float* dither,
                                       const int32_t* aug, int64_t B, int64_t L, int normalize, void* out_btf, int dtype,
                                       int32_t* frames_out, float* pct_out, void* mel_workspace, size_t mel_workspace_bytes) {
  const WaveSrc src = {wave, 0, dither, nullptr, 0ull, 0};
  return set_prefetch(m, src, sample_lens, aug, B, L, normalize, out_btf, dtype, frames_out, pct_out, mel_workspace, mel_workspace_bytes);
}

extern "C" int lasr_model_clear_prefetch(lasr_model_t* m) {
  LASR_CHECK_ARG(m, "lasr_model_clear_prefetch: null model");
  m->prefetch.armed = false;
  return 0;
}

extern "C" int lasr_model_set_prefetch_src(lasr_model_t* m, const lasr_wave_src* wsrc, const int32_t* sample_lens, const int32_t* aug,
                                           int64_t B, int64_t L, int normalize, void* out_btf, int dtype, int32_t* frames_out,
                                           float* pct_out, void* mel_workspace, size_t mel_workspace_bytes) {
  WaveSrc src;
  LASR_TRY(wave_src_from_c(wsrc, &src, "lasr_model_set_prefetch_src"));
  return set_prefetch(m, src, sample_lens, aug, B, L, normalize, out_btf, dtype, frames_out, pct_out, mel_workspace, mel_workspace_bytes);
}

extern "C" int64_t lasr_model_tensor_info(const lasr_model_t* m, int64_t i, char* name, size_t name_cap, int64_t shape[4],
                                          int32_t* ndim, int32_t* kind, int64_t* offset) {
  if (!m) return -1;
  if (i >= 0 && i < (int64_t)m->tensors.size()) {
    const TensorInfo& t = m->tensors[i];
    if (name && name_cap) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (shape) for (int d = 0; d < 4; ++d) shape[d] = t.shape[d];
    if (ndim) *ndim = t.ndim;
    if (kind) *kind = t.kind;
    if (offset) *offset = t.offset;
  }
  return (int64_t)m->tensors.size();
}
extern "C" int64_t lasr_model_param_elems(const lasr_model_t* m) { return m ? m->n_param : -1; }
extern "C" int64_t lasr_model_buffer_elems(const lasr_model_t* m) { return m ? m->n_buffer : -1; }
extern "C" int64_t lasr_model_out_frames(const lasr_model_t* m, int64_t T_in) { (void)m; return out_frames(T_in); }

extern "C" size_t lasr_model_workspace_bytes(lasr_model_t* m, int64_t B, int64_t T_in, int64_t S_max) {
  if (!m || B <= 0 || T_in <= 0) return 0;
  make_plan(m, B, T_in, S_max < 1 ? 1 : S_max);
  return m->plan.total;
}

extern "C" int64_t lasr_model_tap(lasr_model_t* m, const char* name, int64_t B, int64_t T_in, int64_t S_max, int64_t shape[3]) {
  if (!m || !name) return -1;
  make_plan(m, B, T_in, S_max < 1 ? 1 : S_max);
  const Plan& p = m->plan;
  const std::string n(name);
  for (const Unit& u : m->units) {
    if (n == u.tap) { shape[0] = B; shape[1] = p.T; shape[2] = u.co; return (int64_t)u.o_out; }
    if (n == u.tap + ".y") { shape[0] = B; shape[1] = p.T; shape[2] = u.co; return (int64_t)u.o_y; }
    if (n == u.tap + ".y2" && u.has_res) { shape[0] = B; shape[1] = p.T; shape[2] = u.co; return (int64_t)u.o_y2; }
    if (n == u.tap + ".u" && u.has_dw) { shape[0] = B; shape[1] = p.T; shape[2] = u.ci; return (int64_t)u.o_u; }
    if (n == u.tap + ".se_hidden" && u.has_se) { shape[0] = B; shape[1] = 1; shape[2] = u.co / 8; return (int64_t)u.o_se_hid; }   // f32: relu(W1 pooled)
  }
  // gradient ping-pong buffers of the (staged) backward: after a stage that ended at unit i, "bwd.g_cur" holds d(input of unit i)
  // and "bwd.g_prev" still holds d(output of unit i); both are [N][c] tensors at the head of an [N][cmax] allocation
  if (n == "bwd.g_cur" || n == "bwd.g_prev") {
    int64_t cmax = 0;
    for (const Unit& u : m->units) cmax = std::max<int64_t>(cmax, std::max(u.ci, u.co));
    shape[0] = B; shape[1] = p.T; shape[2] = cmax;
    return (int64_t)p.o_g[n == "bwd.g_cur" ? m->bwd_cur : (m->bwd_cur ^ 1)];
  }
  if (n == "ctx_in" && m->cfg.variant != LASR_VARIANT_PLAIN) { shape[0] = B; shape[1] = p.T; shape[2] = 336; return (int64_t)p.o_cat; }
  if (n == "logits") { shape[0] = B; shape[1] = p.T; shape[2] = m->cfg.n_class; return (int64_t)p.o_logits; }
  if (n == "grad_logits") { shape[0] = B; shape[1] = p.T; shape[2] = m->cfg.n_class; return (int64_t)p.o_glogits; }
  if (n == "lens") { shape[0] = B; shape[1] = 1; shape[2] = 1; return (int64_t)p.o_lens; }
  // large-vocabulary head: bf16 logits / gradient with rows padded to 8 classes, and the per-row log-sum-exp
  if (n == "logits_bf16" && lean_capable(m)) { shape[0] = B; shape[1] = p.T; shape[2] = lean_ldc(m); return (int64_t)p.o_logits; }
  if (n == "grad_logits_bf16" && lean_capable(m)) { shape[0] = B; shape[1] = p.T; shape[2] = lean_ldc(m); return (int64_t)p.o_d1; }
  if (n == "lse" && lean_capable(m)) {
    shape[0] = B; shape[1] = p.T; shape[2] = 1;
    return (int64_t)(p.o_lean + align_up(lasr_ctc_workspace_bytes(B, p.T, p.S_max), 256));
  }
  return -1;
}

// bench.py's per-class table: one event pair around a group of consecutive launches of one class (off unless lasr_prof_enable(1))
struct ProfScope {
  int tok; hipStream_t st;
  ProfScope(int kind, void* stream, double bytes) : tok(prof_begin(kind, as_stream(stream), 0.0, bytes)), st(as_stream(stream)) {}
  ~ProfScope() { prof_end(tok, st); }
};

static inline char* at(void* ws, size_t off) { return reinterpret_cast<char*>(ws) + off; }
static inline float* atf(void* ws, size_t off) { return reinterpret_cast<float*>(at(ws, off)); }

// weights as the GEMM's B operand: f32 params directly, or the bf16 shadow copy
static inline const void* wptr(const lasr_model* m, const float* params, void* ws, int64_t off) {
  if (m->cfg.dtype == LASR_BF16) return reinterpret_cast<const bf16_t*>(at(ws, m->plan.o_wbf16)) + off;
  return params + off;
}

// The step's depthwise tap tables (fused.h): made by the training forward's first launch, looked up by the depthwise host
// functions while a model call is in progress on this thread.
struct DwTapScope {
  DwTapCtx ctx;
  bool on = false;
  DwTapScope(const lasr_model* m, const float* params, void* ws, bool valid) {
    ctx.n = 0;
    if (!valid) return;
    for (const Unit& u : m->units)
      if (u.o_taps && ctx.n < 16) {
        ctx.w[ctx.n] = params + u.w_dw; ctx.t[ctx.n] = reinterpret_cast<const uint32_t*>(at(ws, u.o_taps)); ctx.C[ctx.n] = u.ci;
        ++ctx.n;
      }
    if (ctx.n) { dw_taps_set_ctx(&ctx); on = true; }
  }
  ~DwTapScope() { if (on) dw_taps_set_ctx(nullptr); }
  DwTapScope(const DwTapScope&) = delete;
  DwTapScope& operator=(const DwTapScope&) = delete;
};

// eval-mode folding (LASR_NO_EVAL_FOLD=1: the unfolded eval path, for A/B runs and the parity test)
static bool eval_fold(int dtype) {
  static const bool off = getenv("LASR_NO_EVAL_FOLD") != nullptr;
  return !off && dtype == LASR_BF16;
}
static bool fold_unit(const Unit& u) { return u.has_res && u.has_dw && !u.has_se && !u.ctx_before && u.ci % 64 == 0 && u.co % 8 == 0; }

static int forward_impl(lasr_model_t* m, const float* params, float* buffers, const void* feats, const float* pct,
                        int64_t B, int64_t T_in, int training, float* logp_out, int32_t* argmax_out, void* ws,
                        size_t ws_bytes, void* stream, bool lean);

extern "C" int lasr_model_forward(lasr_model_t* m, const float* params, float* buffers, const void* feats, const float* pct,
                                  int64_t B, int64_t T_in, int training, float* logp_out, int32_t* argmax_out, void* ws,
                                  size_t ws_bytes, void* stream) {
  LASR_CHECK_ARG(logp_out, "lasr_model_forward: null pointer");
  return forward_impl(m, params, buffers, feats, pct, B, T_in, training, logp_out, argmax_out, ws, ws_bytes, stream, false);
}

static int forward_impl(lasr_model_t* m, const float* params, float* buffers, const void* feats, const float* pct,
                        int64_t B, int64_t T_in, int training, float* logp_out, int32_t* argmax_out, void* ws,
                        size_t ws_bytes, void* stream, bool lean) {
  LASR_CHECK_ARG(m && params && buffers && feats && pct && (logp_out || lean) && ws, "lasr_model_forward: null pointer");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_in >= 33, "lasr_model_forward: B=%lld T_in=%lld", (long long)B, (long long)T_in);
  make_plan(m, B, T_in, m->planned ? m->plan.S_max : 1);
  const Plan& p = m->plan;
  if (ws_bytes < p.total) return fail(LASR_E_WORKSPACE, "lasr_model_forward: workspace %zu < %zu", ws_bytes, p.total);
  const int dt = m->cfg.dtype;
  const int64_t T = p.T, N = B * T;
  int32_t* lens = reinterpret_cast<int32_t*>(at(ws, p.o_lens));
  const bool dropping = training && m->drop_p > 0.f;
  {
    ProfScope ps(LASR_PROF_OTHER, stream, dt == LASR_BF16 ? 6.0 * m->n_param : 0.0);
    int merged = 1;      // bf16: the lengths and the weights' bf16 shadow in one launch (two independent launch-floor kernels)
    m->taps_valid = false;
    if (dt == LASR_BF16) {
      DwTapJobs jobs;                                       // the depthwise tap tables ride in the same launch (training and eval forwards)
      jobs.n = 0; jobs.blk0[0] = 0;
      {
        for (const Unit& u : m->units)
          if (u.o_taps && jobs.n < 16) {
            jobs.w[jobs.n] = params + u.w_dw; jobs.out[jobs.n] = reinterpret_cast<uint32_t*>(at(ws, u.o_taps)); jobs.C[jobs.n] = u.ci; jobs.k[jobs.n] = u.k;
            jobs.blk0[jobs.n + 1] = jobs.blk0[jobs.n] + (int)cdiv((int64_t)2 * u.ci * kDwTapRow, 256);
            ++jobs.n;
          }
      }
      merged = mask_lengths_step_cast(pct, B, T, lens, dropping ? m->drop_step : nullptr, params, at(ws, p.o_wbf16), m->n_param,
                                      jobs.n ? &jobs : nullptr, stream);
      if (merged < 0 || merged > 1) return merged;
      m->taps_valid = merged == 0 && jobs.n > 0;
    }
    if (merged == 1) {
      LASR_TRY(lasr_mask_lengths_step(pct, B, T, lens, dropping ? m->drop_step : nullptr, stream));   // (bumps the masks' step counter)
      if (dt == LASR_BF16) LASR_TRY(lasr_cast_f32_to_bf16(params, at(ws, p.o_wbf16), m->n_param, stream));
    }
  }
  DwTapScope tap_scope(m, params, ws, m->taps_valid);
  void* scratch = at(ws, p.o_scratch);
  if (!training) {   // eval: BN coefficients of all layers from the running statistics, one launch
    std::vector<lasr_bn_eval_desc> descs;
    for (const Unit& u : m->units) {
      descs.push_back({params + u.bn.gamma, params + u.bn.beta, buffers + u.bn.rmean, buffers + u.bn.rvar, atf(ws, u.o_coef), u.co});
      if (u.has_res)
        descs.push_back({params + u.bn_res.gamma, params + u.bn_res.beta, buffers + u.bn_res.rmean, buffers + u.bn_res.rvar,
                         atf(ws, u.o_coef2), u.co});
    }
    for (size_t i = 0; i < descs.size(); i += 64)
      LASR_TRY(lasr_bn_eval_coef_many(descs.data() + i, (int)std::min<size_t>(64, descs.size() - i), kBnEps, stream));
    if (eval_fold(m->cfg.dtype)) {   // BN folded into the residual units' 1x1 convs: one launch for all of them
      std::vector<lasr_fold_desc> fd;
      for (const Unit& u : m->units)
        if (fold_unit(u))
          fd.push_back({params + u.w_pw, params + u.w_res, atf(ws, u.o_coef), atf(ws, u.o_coef2), at(ws, u.o_wfold), atf(ws, u.o_bfold), u.co, u.ci});
      for (size_t i = 0; i < fd.size(); i += 32)
        LASR_TRY(lasr_fold_bn_weights_many(fd.data() + i, (int)std::min<size_t>(32, fd.size() - i), stream));
    }
  }
  const void* x = feats;
  int64_t Tx = T_in;
  // The BatchNorm + add + activation pass of a plain / residual unit is made by the depthwise launch of the unit above it (fused.h):
  // `pend` is the unit whose pass is still owed.  LASR_BN_DW_FUSE=0: always its own launch (A/B runs; the results are bit-identical).
  static const bool bn_dw_fuse = !(getenv("LASR_BN_DW_FUSE") && atoi(getenv("LASR_BN_DW_FUSE")) == 0);
  const Unit* pend = nullptr;
  auto bn_act_of = [&](const Unit& v) -> int {
    ProfScope ps_bn(LASR_PROF_BN, stream, (double)N * v.co * dtype_size(dt) * (v.has_res ? 3 : 2));
    const lasr_dropout drop = {m->drop_step, m->drop_seed, (uint32_t)(&v - m->units.data()), dropping ? m->drop_p : 0.f};
    return lasr_bn_act_fwd_drop(at(ws, v.o_y), atf(ws, v.o_coef), v.has_res ? at(ws, v.o_y2) : nullptr,
                                v.has_res ? atf(ws, v.o_coef2) : nullptr, v.has_se ? atf(ws, v.o_se_scale) : nullptr, at(ws, v.o_out), dt, B,
                                T, v.co, v.act ? m->cfg.act : LASR_ACT_NONE, dropping ? &drop : nullptr, stream);
  };
  RoctxRange rr_fwd(training ? "lasr:forward (training)" : "lasr:forward (eval)");
  for (size_t ui = 0; ui < m->units.size(); ++ui) {
    const Unit& u = m->units[ui];
    RoctxRange rr_unit(u.rx_fwd.c_str());
    if (pend && !(u.has_dw && u.stride == 1 && !u.ctx_before)) {   // (not reached: the hand-over is only arranged for such a unit)
      LASR_TRY(bn_act_of(*pend));
      pend = nullptr;
    }
    if (u.ctx_before) {
      // context branch: gates' input projection as two GEMMs (f32 out), the recurrence, then cat(x, lstm) -> [N][336]
      {
        lasr_gemm_problem pr[2];
        for (int d = 0; d < 2; ++d) pr[d] = {x, wptr(m, params, ws, m->lstm.w_ih[d]), atf(ws, p.o_gx[d]), N, 160, 256, nullptr, nullptr, 0, nullptr};
        LASR_TRY(lasr_gemm_batch(pr, 2, dt, LASR_F32, 0, 0, 1, scratch, p.scratch_bytes, stream));   // (one launch)
      }
      LASR_TRY(lasr_copy_cols(x, dt, 256, 0, at(ws, p.o_cat), dt, 336, 0, N, 256, 0, stream));
      LASR_TRY(lasr_bilstm_fwd(atf(ws, p.o_gx[0]), atf(ws, p.o_gx[1]), params + m->lstm.w_hh[0], params + m->lstm.w_hh[1],
                               params + m->lstm.b_ih[0], params + m->lstm.b_hh[0], params + m->lstm.b_ih[1], params + m->lstm.b_hh[1],
                               lens, B, T, at(ws, p.o_cat), dt, 336, 256, atf(ws, p.o_lstm_saved), stream));
      x = at(ws, p.o_cat);
    }
    const void* gin = x;
    if (u.has_dw) {
      int fused = 1;
      if (pend) {
        fused = dwconv_fwd_bn(at(ws, pend->o_y), atf(ws, pend->o_coef), pend->has_res ? at(ws, pend->o_y2) : nullptr,
                              pend->has_res ? atf(ws, pend->o_coef2) : nullptr, pend->act ? m->cfg.act : LASR_ACT_NONE, params + u.w_dw,
                              at(ws, pend->o_out), at(ws, u.o_u), B, T, u.ci, u.k, stream);
        if (fused < 0) return fused;
        if (fused == 1) LASR_TRY(bn_act_of(*pend));     // a shape without the fused kernel: the two launches
        pend = nullptr;
      }
      if (fused == 1) LASR_TRY(lasr_dwconv_fwd(x, params + u.w_dw, nullptr, at(ws, u.o_u), dt, B, Tx, u.ci, u.k, u.stride, 0, stream));
      gin = at(ws, u.o_u);
    }
    if (!training && eval_fold(dt) && fold_unit(u)) {
      // out = act([mask(u) | x] . [a W | a2 Wr]^T + (b + b2)): the unit's two 1x1 convs, both BatchNorms, the residual add and
      // the activation as ONE GEMM (no y / y2 round trip, no BN pass)
      LASR_TRY(lasr_gemm_dual(gin, u.ci, x, u.ci, at(ws, u.o_wfold), atf(ws, u.o_bfold), at(ws, u.o_out), N, u.co, u.masked ? lens : nullptr,
                              T, u.act ? m->cfg.act : LASR_ACT_NONE, stream));
      x = at(ws, u.o_out);
      Tx = T;
      continue;
    }
    float* stats = training ? atf(ws, u.o_stats) : nullptr;
    float* stats2 = (training && u.has_res) ? atf(ws, u.o_stats2) : nullptr;
    bool se_sums_done = false;
    {  // main 1x1 (masked, BN sums) and, for residual blocks, the residual 1x1 (never masked) in one launch
      lasr_gemm_problem pr[2];
      pr[0] = {gin, wptr(m, params, ws, u.w_pw), at(ws, u.o_y), N, u.co, u.ci, nullptr, u.masked ? lens : nullptr, T, stats};
      if (u.has_res) pr[1] = {x, wptr(m, params, ws, u.w_res), at(ws, u.o_y2), N, u.co, u.ci, nullptr, nullptr, 0, stats2};
      const int np = u.has_res ? 2 : 1;
      if (training && !no_fuse()) {
        // the epilogue's per-tile BN sums go straight to ONE reduce+finalize launch for both branches
        const float* parts[2] = {nullptr, nullptr};
        int tiles[2] = {0, 0};
        LASR_TRY(lasr_gemm_batch_partials(pr, np, dt, dt, 0, 0, scratch, p.scratch_bytes, parts, tiles, stream));
        lasr_bn_branch br[2];
        br[0] = {parts[0], tiles[0], params + u.bn.gamma, params + u.bn.beta, buffers + u.bn.rmean, buffers + u.bn.rvar,
                 atf(ws, u.o_coef), atf(ws, u.o_saved), stats};
        if (u.has_res)
          br[1] = {parts[1], tiles[1], params + u.bn_res.gamma, params + u.bn_res.beta, buffers + u.bn_res.rmean,
                   buffers + u.bn_res.rvar, atf(ws, u.o_coef2), atf(ws, u.o_saved2), stats2};
        ProfScope ps(LASR_PROF_BN, stream, 0.0);
        // SE units: the squeeze's per-utterance sums of y do not depend on the finalize - one launch for both (se_seqsum.h)
        int merged = 1;
        if (u.has_se) {
          merged = bn_finalize_partials_seqsum(br, np, u.co, N, kBnEps, kBnMom, at(ws, u.o_y), dt, B, T, atf(ws, u.o_se_sum), stream);
          if (merged < 0 || merged > 1) return merged;
          se_sums_done = merged == 0;
        }
        if (merged == 1) LASR_TRY(lasr_bn_finalize_partials(br, np, u.co, N, kBnEps, kBnMom, stream));
      } else {
        LASR_TRY(lasr_gemm_batch(pr, np, dt, dt, 0, 0, 1, scratch, p.scratch_bytes, stream));
        if (training) {   // (eval: every layer's coefficients were computed by one launch before the loop)
          LASR_TRY(lasr_bn_finalize(stats, params + u.bn.gamma, params + u.bn.beta, buffers + u.bn.rmean, buffers + u.bn.rvar,
                                    atf(ws, u.o_coef), atf(ws, u.o_saved), u.co, N, kBnEps, kBnMom, training, stream));
          if (u.has_res)
            LASR_TRY(lasr_bn_finalize(stats2, params + u.bn_res.gamma, params + u.bn_res.beta, buffers + u.bn_res.rmean,
                                      buffers + u.bn_res.rvar, atf(ws, u.o_coef2), atf(ws, u.o_saved2), u.co, N, kBnEps, kBnMom,
                                      training, stream));
        }
      }
    }
    bool se_applied = false;
    if (u.has_se) {  // squeeze over all T' frames of BN(y) (affine in the per-utterance sums of y), excite MLP
      ProfScope ps_se(LASR_PROF_BN, stream, (double)N * u.co * dtype_size(dt));
      if (!se_sums_done) LASR_TRY(lasr_seqsum(at(ws, u.o_y), dt, B, T, u.co, atf(ws, u.o_se_sum), stream));
      int folded = 1;
      if (training && !dropping && !no_fuse())   // round 5: the excite MLP inside the apply pass (one launch instead of three)
        folded = bn_se_act_fwd(at(ws, u.o_y), atf(ws, u.o_coef), u.has_res ? at(ws, u.o_y2) : nullptr, u.has_res ? atf(ws, u.o_coef2) : nullptr,
                               atf(ws, u.o_se_sum), params + u.w_se1, params + u.w_se2, at(ws, u.o_out), atf(ws, u.o_se_pool),
                               atf(ws, u.o_se_hid), atf(ws, u.o_se_scale), dt, B, T, u.co, u.act ? m->cfg.act : LASR_ACT_NONE, stream);
      if (folded < 0 || folded > 1) return folded;
      se_applied = folded == 0;
      if (!se_applied)
        LASR_TRY(lasr_se_fwd(atf(ws, u.o_se_sum), atf(ws, u.o_coef), params + u.w_se1, params + u.w_se2, B, T, u.co, atf(ws, u.o_se_pool),
                             atf(ws, u.o_se_hid), atf(ws, u.o_se_scale), stream));
    }
    // BN + residual add + activation: by the next unit's depthwise launch when that is a stride-1 bf16 depthwise conv reading this
    // unit's output directly (training; no SE scale, no dropout mask in between), otherwise here
    const Unit* nx = ui + 1 < m->units.size() ? &m->units[ui + 1] : nullptr;
    if (bn_dw_fuse && training && !no_fuse() && dt == LASR_BF16 && !u.has_se && !dropping && nx && nx->has_dw && nx->stride == 1 &&
        !nx->ctx_before && nx->ci == u.co)
      pend = &u;
    else if (!se_applied)
      LASR_TRY(bn_act_of(u));
    x = at(ws, u.o_out);
    Tx = T;
  }
  if (pend) LASR_TRY(bn_act_of(*pend));   // (not reached: the last unit has no unit above it)
  // decoder 1x1 1024 -> C with bias (models/QuartNet.py:275), f32 logits, then log_softmax (+argmax)
  const int64_t C = m->cfg.n_class;
  // a narrow vocabulary leaves N/128 = 1 tile column: split K so that the 32 MB of activations are streamed by
  // 4 x 126 workgroups instead of 126 (the split partials are C/1024 of the input: negligible)
  if (lean) {   // bf16 logits + per-tile softmax statistics; lse / emissions / gradient follow in lasr_ctc_loss_lean
    float* rs = atf(ws, p.o_rowstat);
    int32_t* ra = reinterpret_cast<int32_t*>(rs + (size_t)N * cdiv(C, 256) * 2);
    return lasr_gemm_rowstat(x, wptr(m, params, ws, m->w_dec), params + m->b_dec, at(ws, p.o_logits), lean_ldc(m), N, C, 1024, rs, ra,
                             &m->lean_tiles, stream);
  }
  const int dec_split = (dt == LASR_BF16 && C <= 128) ? dec_split_k() : 1;
  if (dec_split > 1 && log_softmax_split_on(C)) {      // narrow head: the split-K slabs are summed by the log_softmax launch itself (fused.h)
    int splits = 0;
    LASR_TRY(gemm_split_partials_one(x, wptr(m, params, ws, m->w_dec), dt, N, C, 1024, 0, 0, dec_split, scratch, p.scratch_bytes, &splits, stream));
    ProfScope ps(LASR_PROF_HEAD, stream, (2.0 + splits) * N * C * sizeof(float));
    const int rc = log_softmax_split(reinterpret_cast<const float*>(scratch), splits, params + m->b_dec, atf(ws, p.o_logits), logp_out, argmax_out,
                                     N, C, stream);
    return rc;
  }
  LASR_TRY(lasr_gemm(x, wptr(m, params, ws, m->w_dec), atf(ws, p.o_logits), dt, LASR_F32, N, C, 1024, 0, 0, params + m->b_dec,
                     nullptr, nullptr, 0, nullptr, dec_split, scratch, p.scratch_bytes, stream));
  ProfScope ps(LASR_PROF_HEAD, stream, 2.0 * N * C * sizeof(float));
  LASR_TRY(lasr_log_softmax(atf(ws, p.o_logits), logp_out, argmax_out, N, C, stream));
  return 0;
}

// The 1x1 weight gradients collected over a backward stage, as ONE split-K launch.  The slice count is chosen for
// ~3 rounds of resident workgroups (512 at two per CU): a single unit needs 16 slices for that, a stage of 6-13 units
// needs 2-6, so every workgroup runs 3-8x more K steps between its prologue and its slab write-out.
static int flush_wgrads(std::vector<lasr_gemm_problem>& probs, std::vector<float*>& slabs, std::vector<lasr_reduce_desc>& pending,
                        void* stream) {
  const int split = wgrad_split();   // the cap the slabs were sized for; the library picks the slice count for its tile form
  int splits[32];
  // round 5: whatever `pending` holds at this point is complete (the depthwise weight gradients' per-utterance partials, the BiLSTM's) -
  // the launch's one round of tiles leaves CUs idle (71 tiles x 3 slices = 213 of 256 at cfg2), and extra workgroups of the same launch
  // do those reductions there; what they took leaves the list lasr_reduce_many gets (123 -> 59 MB at cfg2).  Same sums, same order.
  int taken = 0;
  LASR_TRY(gemm_multi_split_partials_riders(probs.data(), (int)probs.size(), split, slabs.data(), splits, pending.data(), (int)pending.size(),
                                            &taken, stream));
  if (taken > 0) pending.erase(pending.begin(), pending.begin() + taken);
  for (size_t i = 0; i < probs.size(); ++i)
    pending.push_back({slabs[i], reinterpret_cast<float*>(probs[i].C), probs[i].M * probs[i].N, splits[i]});
  probs.clear();
  slabs.clear();
  return 0;
}

// Split-K slices of the decoder's weight gradient dW = gl^T h ([C][1024], K = N rows) on the 128 x 128 tile (two workgroups
// per CU): rounds of resident workgroups x K steps per slice, against the f32 slabs every slice writes and the reduction reads
// back.  C = 28: 8 tiles, the cap of 16 slices (as before).  C = 4334: 272 tiles, where 16 slices meant 9 rounds and 284 MB of
// slabs (214 us + an 84 us reduction at cfg5); the model's minimum is ~5 slices.
static int dec_wgrad_split(int64_t C, int64_t N) {
  const int64_t tiles = cdiv(C, 128) * 8;
  const double t_k = 15.8e-9;                                     // one K element of one tile at ~1.06 PFLOP/s over 512 workgroups
  const double t_slab = (double)C * 1024 * 8 / 4e12;              // one slab written and read back at ~4 TB/s
  int best = 1;
  double best_t = 1e30;
  for (int sp = 1; sp <= 16; ++sp) {
    const double rounds = (double)cdiv(tiles * sp, 512);
    const double t = rounds * (double)cdiv(N, sp) * t_k + (sp > 1 ? sp * t_slab : 0.0);
    if (t < best_t) { best_t = t; best = sp; }
  }
  return best;
}

// backward from d(loss)/d(logits) already in the workspace (o_glogits)
// unit_stop: the unit loop runs from the last unit down to `unit_stop` (0 = the whole model); a later
// lasr_model_backward_continue call picks up at unit_stop-1.  with_head: run the decoder part first.
static int backward_from_glogits(lasr_model* m, const float* params, const void* feats, int64_t B, int64_t T_in, float* grads,
                                 void* ws, void* stream, bool with_head = true, int unit_hi = -1, int unit_stop = 0) {
  const Plan& p = m->plan;
  const int dt = m->cfg.dtype;
  const int64_t T = p.T, N = B * T, C = m->cfg.n_class;
  DwTapScope tap_scope(m, params, ws, m->taps_valid);     // (the tables of this step's forward: the parameters have not moved since)
  void* scratch = at(ws, p.o_scratch);
  const size_t sb = p.scratch_bytes;
  const int32_t* lens = reinterpret_cast<const int32_t*>(at(ws, p.o_lens));
  float* gl = atf(ws, p.o_glogits);
  const Unit& last = m->units.back();
  int cur = m->bwd_cur;
  // the batch-mean loss forward_and_loss left to this call's first launch: taken over HERE and forgotten by the model, so that an
  // error return below (or a later, separate lasr_model_backward) can never write through a stale caller pointer
  const float* tail_nll = m->tail_nll;
  float* tail_loss = m->tail_loss;
  m->tail_nll = nullptr; m->tail_loss = nullptr;
  if (with_head) {
  cur = 0;
  // decoder (models/QuartNet.py:275): dW = gl^T h, db = colsum(gl), dh = gl W
  const void* gl_ab = gl;
  int64_t ld_gl = C;
  bool bias_done = false;
  if (m->lean_active) {   // lasr_ctc_loss_lean wrote the bf16 gradient (rows padded to 8) and the bias gradient
    ld_gl = lean_ldc(m);
    gl_ab = at(ws, p.o_d1);
  } else if (dt == LASR_BF16) {  // GEMM operands share a dtype: bf16 shadow of the logits gradient, rows padded to 16-byte multiples
    ld_gl = (C + 7) / 8 * 8;
    ProfScope ps(LASR_PROF_HEAD, stream, (double)N * C * 6);
    int merged = 1;
    if (tail_loss) {         // padded copy + bias-gradient column sums + the loss mean forward_and_loss left to this launch
      merged = head_tail(gl, N, C, at(ws, p.o_d1), ld_gl, grads + m->b_dec, scratch, sb, tail_nll, B, 1.0f / (float)B, tail_loss, stream);
      if (merged < 0 || merged > 1) return merged;
      if (merged == 1) LASR_TRY(lasr_scale_sum_f32(tail_nll, B, 1.0f / (float)B, tail_loss, stream));
      tail_loss = nullptr;
      bias_done = merged == 0;
    }
    if (merged == 1) LASR_TRY(lasr_cast_pad_f32_to_bf16(gl, at(ws, p.o_d1), N, C, ld_gl, stream));
    gl_ab = at(ws, p.o_d1);
  }
  if (tail_loss) {           // (not reached in bf16; keeps the loss defined whatever path the head takes)
    LASR_TRY(lasr_scale_sum_f32(tail_nll, B, 1.0f / (float)B, tail_loss, stream));
    tail_loss = nullptr;
  }
  LASR_TRY(lasr_gemm_ld(gl_ab, ld_gl, at(ws, last.o_out), 1024, grads + m->w_dec, 1024, dt, LASR_F32, C, 1024, N, 1, 1, nullptr,
                        dec_wgrad_split(C, N), scratch, sb, stream));
  if (!m->lean_active && !bias_done) {
    ProfScope ps(LASR_PROF_HEAD, stream, (double)N * C * 4);
    LASR_TRY(lasr_colsum_f32(gl, grads + m->b_dec, N, C, scratch, sb, stream));
  }
  LASR_TRY(lasr_gemm_ld(gl_ab, ld_gl, wptr(m, params, ws, m->w_dec), 1024, at(ws, p.o_g[cur]), 1024, dt, dt, N, 1024, C, 0, 1, nullptr,
                        1, scratch, sb, stream));
  }
  if (unit_hi < 0) unit_hi = (int)m->units.size() - 1;
  // The "sum the partial slabs" tails of the weight-gradient kernels are collected and issued as ONE launch when
  // the stage ends (27 launches of ~5 us each off the step; LASR_NO_FUSE=1 keeps them inline for A/B runs).
  static const bool no_defer = getenv("LASR_NO_DEFER") != nullptr;
  const bool defer = !no_fuse() && !no_defer;
  std::vector<lasr_reduce_desc> pending;
  std::vector<lasr_gemm_problem> wprobs;
  std::vector<float*> wslabs;
  RoctxRange rr_bwd("lasr:backward");
  for (int ui = unit_hi; ui >= unit_stop; --ui) {
    RoctxRange rr_unit(m->units[ui].rx_bwd.c_str());
    const Unit& u = m->units[ui];
    const void* x_in = ui > 0 ? at(ws, m->units[ui - 1].o_out) : feats;
    if (u.ctx_before) x_in = at(ws, p.o_cat);
    const int64_t Tx = ui > 0 ? T : T_in;
    const int act = u.act ? m->cfg.act : LASR_ACT_NONE;
    void* dout = at(ws, p.o_g[cur]);
    // weight-gradient GEMMs batched per stage (the multi-problem kernel loads 16-byte operand rows)
    const bool defer_w = defer && dt == LASR_BF16 && u.co % 8 == 0 && u.ci % 8 == 0;
    void* dy = defer_w ? at(ws, u.o_dy) : at(ws, p.o_d1);
    void* dy2 = u.has_res ? (defer_w ? at(ws, u.o_dy2) : at(ws, p.o_d2)) : nullptr;
    const float* se_scale = u.has_se ? atf(ws, u.o_se_scale) : nullptr;
    const float* se_grad = u.has_se ? atf(ws, u.o_se_grad) : nullptr;
    const lasr_dropout drop = {m->drop_step, m->drop_seed, (uint32_t)ui, m->drop_p};   // the masks of this step's forward
    const lasr_dropout* dq = m->drop_p > 0.f ? &drop : nullptr;
    static const bool se_unfused = getenv("LASR_SE_UNFUSED_BWD") != nullptr;    // A/B switch: the three-pass form
    const bool se_fused = u.has_se && !se_unfused && !no_fuse();
    {   // BatchNorm (+ activation, residual add, SE) backward of the unit: two passes over (dout, y[, y2]), writes dy[, dy2]
    ProfScope ps_bn(LASR_PROF_BN, stream, (double)N * u.co * dtype_size(dt) * (u.has_res ? 8 : 5));
    if (se_fused) {
      // SE units: statistics, SE-scale gradient, excite-MLP backward and BN apply in two passes over (dout, y, y2)
      LASR_TRY(lasr_bn_se_bwd(dout, at(ws, u.o_y), atf(ws, u.o_coef), atf(ws, u.o_saved), params + u.bn.gamma, params + u.bn.beta,
                              u.has_res ? at(ws, u.o_y2) : nullptr, u.has_res ? atf(ws, u.o_coef2) : nullptr,
                              u.has_res ? atf(ws, u.o_saved2) : nullptr, u.has_res ? params + u.bn_res.gamma : nullptr, se_scale,
                              atf(ws, u.o_se_hid), atf(ws, u.o_se_pool), atf(ws, u.o_se_sum), params + u.w_se1, params + u.w_se2,
                              u.masked ? lens : nullptr, dy, dy2, grads + u.bn.gamma, grads + u.bn.beta,
                              u.has_res ? grads + u.bn_res.gamma : nullptr, u.has_res ? grads + u.bn_res.beta : nullptr, grads + u.w_se1,
                              grads + u.w_se2, atf(ws, u.o_se_grad), dt, B, T, u.co, act, dq, scratch, sb, stream));
    } else {
    if (u.has_se)
      LASR_TRY(lasr_se_bwd_drop(dout, at(ws, u.o_y), atf(ws, u.o_coef), u.has_res ? at(ws, u.o_y2) : nullptr,
                                u.has_res ? atf(ws, u.o_coef2) : nullptr, se_scale, atf(ws, u.o_se_hid), atf(ws, u.o_se_pool), params + u.w_se1,
                                params + u.w_se2, dt, B, T, u.co, act, dq, atf(ws, u.o_se_grad), grads + u.w_se1, grads + u.w_se2, scratch, sb,
                                stream));
    // fused hand-over: pass 2 reduces pass 1's partial sums (LASR_NO_FUSE=1 keeps the separate reductions, for A/B runs)
    float* fsum = no_fuse() ? atf(ws, p.o_sums) : nullptr;
    float* fsum2 = no_fuse() ? atf(ws, p.o_sums2) : nullptr;
    LASR_TRY(lasr_bn_act_bwd_stats_drop(dout, at(ws, u.o_y), atf(ws, u.o_coef), atf(ws, u.o_saved), u.has_res ? at(ws, u.o_y2) : nullptr,
                                        u.has_res ? atf(ws, u.o_coef2) : nullptr, u.has_res ? atf(ws, u.o_saved2) : nullptr, se_scale,
                                        se_grad, fsum, fsum2, dt, B, T, u.co, act, dq, scratch, sb, stream));
    LASR_TRY(lasr_bn_act_bwd_apply_drop(dout, at(ws, u.o_y), atf(ws, u.o_coef), atf(ws, u.o_saved), params + u.bn.gamma,
                                        u.has_res ? at(ws, u.o_y2) : nullptr, u.has_res ? atf(ws, u.o_coef2) : nullptr,
                                        u.has_res ? atf(ws, u.o_saved2) : nullptr, u.has_res ? params + u.bn_res.gamma : nullptr, se_scale,
                                        se_grad, fsum, fsum2, u.masked ? lens : nullptr, dy, dy2,
                                        grads + u.bn.gamma, grads + u.bn.beta, u.has_res ? grads + u.bn_res.gamma : nullptr,
                                        u.has_res ? grads + u.bn_res.beta : nullptr, dt, B, T, u.co, act, dq, scratch, sb, stream));
    }
    }
    // weight gradients of the main and residual 1x1: dW[co][ci] = dy^T gin, dWr = dy2^T x  (one split-K launch)
    const void* gin = u.has_dw ? at(ws, u.o_u) : x_in;
    {
      lasr_gemm_problem pr[2];
      pr[0] = {dy, gin, grads + u.w_pw, u.co, u.ci, N, nullptr, nullptr, 0, nullptr};
      if (u.has_res) pr[1] = {dy2, x_in, grads + u.w_res, u.co, u.ci, N, nullptr, nullptr, 0, nullptr};
      if (defer_w) {
        float* slab = atf(ws, u.o_wgp);
        if (wprobs.size() + (u.has_res ? 2 : 1) > 32) LASR_TRY(flush_wgrads(wprobs, wslabs, pending, stream));   // (the launch takes 32 problems)
        wprobs.push_back(pr[0]); wslabs.push_back(slab);
        if (u.has_res) { wprobs.push_back(pr[1]); wslabs.push_back(slab + u.wgp_bytes / (2 * sizeof(float))); }
      } else {
        LASR_TRY(lasr_gemm_batch(pr, u.has_res ? 2 : 1, dt, LASR_F32, 1, 1, wgrad_split(), scratch, sb, stream));
      }
    }
    const bool need_dx = ui > 0;
    void* dx = at(ws, p.o_g[cur ^ 1]);
    if (u.has_dw) {
      // data gradients: d(dw output) = dy Wp and, for residual blocks, dx_res = dy2 Wr (one launch)
      lasr_gemm_problem pr[2];
      pr[0] = {dy, wptr(m, params, ws, u.w_pw), at(ws, p.o_du), N, u.ci, u.co, nullptr, nullptr, 0, nullptr};
      const bool with_res = u.has_res && need_dx;
      if (with_res) pr[1] = {dy2, wptr(m, params, ws, u.w_res), at(ws, p.o_dxr), N, u.ci, u.co, nullptr, nullptr, 0, nullptr};
      LASR_TRY(lasr_gemm_batch(pr, with_res ? 2 : 1, dt, dt, 0, 1, 1, scratch, sb, stream));
      // depthwise dW from (x, du); dx = flipped depthwise conv of du (+ residual dx)
      const bool fused_dw = defer && need_dx && u.stride == 1;   // both in one launch (falls back inside for other dtypes / shapes)
      if (fused_dw) {
        int npart = 0;
        LASR_TRY(lasr_dwconv_bwd_fused(x_in, at(ws, p.o_du), params + u.w_dw, u.has_res ? at(ws, p.o_dxr) : nullptr, dx, dt, B, T, u.ci, u.k,
                                       at(ws, u.o_dwp), u.dwp_bytes, &npart, stream));
        pending.push_back({atf(ws, u.o_dwp), grads + u.w_dw, u.ci * (int64_t)u.k, npart});
      } else if (defer) {
        int npart = 0;
        LASR_TRY(lasr_dwconv_wgrad_partials(x_in, at(ws, p.o_du), dt, B, Tx, u.ci, u.k, u.stride, at(ws, u.o_dwp), u.dwp_bytes, &npart,
                                            stream));
        pending.push_back({atf(ws, u.o_dwp), grads + u.w_dw, u.ci * (int64_t)u.k, npart});
      } else {
        LASR_TRY(lasr_dwconv_wgrad(x_in, at(ws, p.o_du), grads + u.w_dw, dt, B, Tx, u.ci, u.k, u.stride, scratch, sb, stream));
      }
      if (pending.size() + wprobs.size() > 60) {
        if (!wprobs.empty()) LASR_TRY(flush_wgrads(wprobs, wslabs, pending, stream));
        LASR_TRY(lasr_reduce_many(pending.data(), (int)pending.size(), stream));
        pending.clear();
      }
      if (need_dx && !fused_dw)
        LASR_TRY(lasr_dwconv_fwd(at(ws, p.o_du), params + u.w_dw, u.has_res ? at(ws, p.o_dxr) : nullptr, dx, dt, B, T, u.ci, u.k, 1,
                                 1, stream));
    } else if (need_dx) {
      if (u.has_res)
        LASR_TRY(lasr_gemm(dy2, wptr(m, params, ws, u.w_res), at(ws, p.o_dxr), dt, dt, N, u.ci, u.co, 0, 1, nullptr, nullptr, nullptr,
                           0, nullptr, 1, scratch, sb, stream));
      LASR_TRY(lasr_gemm(dy, wptr(m, params, ws, u.w_pw), dx, dt, dt, N, u.ci, u.co, 0, 1, nullptr, u.has_res ? at(ws, p.o_dxr) : nullptr,
                         nullptr, 0, nullptr, 1, scratch, sb, stream));
    }
    cur ^= 1;
    if (u.ctx_before) {
      // g[cur] holds d(cat) [N][336].  Back through the BiLSTM, then
      // d(block23 out) = d(cat)[:, :256] + dG_f W_ih_f + dG_r W_ih_r  -> g[cur^1]
      const void* x23 = at(ws, m->units[ui - 1].o_out);
      // The recurrence (437 us of dependent steps on 2 B small workgroups) next to the 1x1 weight gradients collected so far in this
      // stage - the units above block3 and block3's own: they do not depend on it - in ONE grid (gemm_bf16.hip,
      // gemm_bf16_big_multi_lstm_kernel; LASR_LSTM_BESIDE_WGRAD=0: the two launches one after the other).  `scratch` holds the
      // recurrence's dW_hh partials; the GEMM part writes only its slabs.
      int beside = 1;
      if (!wprobs.empty() && wprobs.size() <= 32 && dt == LASR_BF16 && defer) {
        int splits[32], taken = 0;
        const lstm::BwdArgs rec = {at(ws, p.o_g[cur]), 336, 256, params + m->lstm.w_hh[0], params + m->lstm.w_hh[1], lens, T, atf(ws, p.o_lstm_saved),
                                   atf(ws, p.o_dg[0]), atf(ws, p.o_dg[1]), atf(ws, p.o_lstm_part),
                                   reinterpret_cast<bf16_t*>(at(ws, p.o_dgb[0])), reinterpret_cast<bf16_t*>(at(ws, p.o_dgb[1])), atf(ws, p.o_lstm_bpart)};
        beside = gemm_multi_split_partials_with_bilstm_bwd(wprobs.data(), (int)wprobs.size(), &taken, wgrad_split(), wslabs.data(), splits, rec, dt, B,
                                                           stream);
        if (beside < 0 || beside > 1) return beside;
        if (beside == 0) {                                  // the first `taken` problems rode along; the rest wait for the stage's closing launch
          for (int i = 0; i < taken; ++i)
            pending.push_back({wslabs[i], reinterpret_cast<float*>(wprobs[i].C), wprobs[i].M * wprobs[i].N, splits[i]});
          wprobs.erase(wprobs.begin(), wprobs.begin() + taken);
          wslabs.erase(wslabs.begin(), wslabs.begin() + taken);
        }
      }
      if (beside == 0) {
        // The recurrences' launch left dg as f32 and bf16, and per-workgroup partial sums of dW_hh and of dg's columns.  Their sums
        // (dW_hh, and the bias gradients - b_ih and b_hh enter the gates as a sum: one gradient, stored twice) join the stage's closing
        // reduction; dW_ih = dG^T x23 joins the stage's closing weight-gradient launch; only the data gradient
        // d(block23 out) = d(cat)[:, :256] + dG_f W_ih_f + dG_r W_ih_r is needed now.   (7 small launches fewer than the separate form)
        const int np = (int)B * lstm::kDwZ;
        const int64_t gh = (int64_t)lstm::G * lstm::H;
        if (pending.size() + wprobs.size() + 8 > 64) {      // six segments + two problems are about to join (lasr_reduce_many takes 64)
          LASR_TRY(flush_wgrads(wprobs, wslabs, pending, stream));
          LASR_TRY(lasr_reduce_many(pending.data(), (int)pending.size(), stream));
          pending.clear();
        }
        for (int d = 0; d < 2; ++d) {
          pending.push_back({atf(ws, p.o_lstm_part) + (int64_t)d * np * gh, grads + m->lstm.w_hh[d], gh, np});
          pending.push_back({atf(ws, p.o_lstm_bpart) + (int64_t)d * np * lstm::G, grads + m->lstm.b_ih[d], lstm::G, np});
          pending.push_back({atf(ws, p.o_lstm_bpart) + (int64_t)d * np * lstm::G, grads + m->lstm.b_hh[d], lstm::G, np});
          if (wprobs.size() + 1 > 32) LASR_TRY(flush_wgrads(wprobs, wslabs, pending, stream));
          wprobs.push_back({at(ws, p.o_dgb[d]), x23, grads + m->lstm.w_ih[d], 160, 256, N, nullptr, nullptr, 0, nullptr});
          wslabs.push_back(atf(ws, p.o_lstm_wgp) + (int64_t)d * (p.lstm_wgp_bytes / sizeof(float)));
        }
        if (pending.size() + wprobs.size() > 60) {          // (lasr_reduce_many takes 64 segments)
          LASR_TRY(flush_wgrads(wprobs, wslabs, pending, stream));
          LASR_TRY(lasr_reduce_many(pending.data(), (int)pending.size(), stream));
          pending.clear();
        }
        LASR_TRY(lasr_copy_cols(at(ws, p.o_g[cur]), dt, 336, 0, at(ws, p.o_g[cur ^ 1]), dt, 256, 0, N, 256, 0, stream));
        for (int d = 0; d < 2; ++d)
          LASR_TRY(lasr_gemm(at(ws, p.o_dgb[d]), wptr(m, params, ws, m->lstm.w_ih[d]), at(ws, p.o_g[cur ^ 1]), dt, dt, N, 256, 160, 0, 1, nullptr,
                             at(ws, p.o_g[cur ^ 1]), nullptr, 0, nullptr, 1, scratch, sb, stream));
        cur ^= 1;
        continue;
      }
      LASR_TRY(lasr_bilstm_bwd(at(ws, p.o_g[cur]), dt, 336, 256, params + m->lstm.w_hh[0], params + m->lstm.w_hh[1], lens, B, T,
                               atf(ws, p.o_lstm_saved), atf(ws, p.o_dg[0]), atf(ws, p.o_dg[1]), grads + m->lstm.w_hh[0],
                               grads + m->lstm.w_hh[1], scratch, sb, stream));
      LASR_TRY(lasr_copy_cols(at(ws, p.o_g[cur]), dt, 336, 0, at(ws, p.o_g[cur ^ 1]), dt, 256, 0, N, 256, 0, stream));
      for (int d = 0; d < 2; ++d) {
        const void* dg_ab = atf(ws, p.o_dg[d]);
        if (dt == LASR_BF16) {
          LASR_TRY(lasr_cast_f32_to_bf16(atf(ws, p.o_dg[d]), at(ws, p.o_d1), N * 160, stream));
          dg_ab = at(ws, p.o_d1);
        }
        LASR_TRY(lasr_colsum_f32(atf(ws, p.o_dg[d]), grads + m->lstm.b_ih[d], N, 160, scratch, sb, stream));
        // both biases enter the gates as b_ih + b_hh: one gradient, stored twice
        LASR_TRY(lasr_copy_cols(grads + m->lstm.b_ih[d], LASR_F32, 160, 0, grads + m->lstm.b_hh[d], LASR_F32, 160, 0, 1, 160, 0, stream));
        // (dW_ih = dG^T x23 stays its own split-K launch: as problems 86 and 87 of the stage's batched launch the two one-tile problems
        //  push it from 3 slices x 85 tiles = 255 workgroups to 2 x 87 = 174, +47 us - exactly what the two small launches cost)
        LASR_TRY(lasr_gemm(dg_ab, x23, grads + m->lstm.w_ih[d], dt, LASR_F32, 160, 256, N, 1, 1, nullptr, nullptr, nullptr, 0, nullptr, 16,
                           scratch, sb, stream));
        LASR_TRY(lasr_gemm(dg_ab, wptr(m, params, ws, m->lstm.w_ih[d]), at(ws, p.o_g[cur ^ 1]), dt, dt, N, 256, 160, 0, 1, nullptr,
                           at(ws, p.o_g[cur ^ 1]), nullptr, 0, nullptr, 1, scratch, sb, stream));
      }
      cur ^= 1;
    }
  }
  RoctxRange rr_tail("lasr:wgrad flush + reduce");
  if (!wprobs.empty()) LASR_TRY(flush_wgrads(wprobs, wslabs, pending, stream));
  if (!pending.empty()) {   // the stage's gradients are final
    double rb = 0;
    for (const lasr_reduce_desc& d : pending) rb += (double)d.n * (d.n_partials + 1) * sizeof(float);
    ProfScope ps(LASR_PROF_OTHER, stream, rb);
    LASR_TRY(lasr_reduce_many(pending.data(), (int)pending.size(), stream));
  }
  m->bwd_cur = cur;
  m->bwd_next = unit_stop - 1;
  return 0;
}

extern "C" int lasr_model_backward(lasr_model_t* m, const float* params, const void* feats, const float* logp,
                                   const float* grad_logp, int64_t B, int64_t T_in, float* grads, void* ws, size_t ws_bytes,
                                   void* stream) {
  LASR_CHECK_ARG(m && params && feats && logp && grad_logp && grads && ws, "lasr_model_backward: null pointer");
  LASR_CHECK_ARG(m->planned && m->plan.B == B && m->plan.T_in == T_in, "lasr_model_backward: no matching forward in this workspace");
  if (ws_bytes < m->plan.total) return fail(LASR_E_WORKSPACE, "lasr_model_backward: workspace");
  const int64_t N = B * m->plan.T;
  m->lean_active = false;
  LASR_TRY(lasr_log_softmax_bwd(logp, grad_logp, atf(ws, m->plan.o_glogits), N, m->cfg.n_class, stream));
  return backward_from_glogits(m, params, feats, B, T_in, grads, ws, stream);
}

// forward + mean CTC + d(loss)/d(logits): the dense head (f32 logits, log_softmax, lasr_ctc_loss[_mel]) or, for a large
// vocabulary in bf16 when the caller does not ask for the log-probs (logp_out == NULL), the lean head of ctc_lean.hip
static int forward_and_loss(lasr_model_t* m, const float* params, float* buffers, const void* feats, const float* pct,
                            const int64_t* targets, const int32_t* tgt_lens, int64_t B, int64_t T_in, int64_t S_max, float* logp_out,
                            float* loss_out, float* nll_out, int32_t* argmax_out, float* grads, void* ws, size_t ws_bytes, void* stream,
                            const char* who) {
  LASR_CHECK_ARG(m && targets && tgt_lens && loss_out && nll_out && grads, "%s: null pointer", who);
  make_plan(m, B, T_in, S_max < 1 ? 1 : S_max);
  const Plan& p = m->plan;
  if (ws_bytes < p.total) return fail(LASR_E_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, p.total);
  const bool lean = logp_out == nullptr;
  LASR_CHECK_ARG(!lean || lean_capable(m), "%s: logp_out may be NULL only for the large-vocabulary bf16 head (C >= 256)", who);
  m->lean_active = lean;
  LASR_TRY(forward_impl(m, params, buffers, feats, pct, B, T_in, 1, logp_out, argmax_out, ws, ws_bytes, stream, lean));
  const int C = m->cfg.n_class;
  const int32_t* lens = reinterpret_cast<const int32_t*>(at(ws, p.o_lens));
  // mean_b CTC(blank = C-1) with lengths int(T'*pct) (train.py:76-78); its gradient w.r.t. the
  // log-probs is (softmax - occupancy)/B, which log_softmax backward maps to itself.
  const lasr_model::Prefetch pf = m->prefetch;
  m->prefetch.armed = false;
  // loss head: the lattice reads the log-probs and writes their gradient (dense: 2 x N x C f32; lean: bf16 logits in, bf16 gradient out)
  ProfScope ps_head(LASR_PROF_HEAD, stream, lean ? 2.0 * B * p.T * lean_ldc(m) * 2 : 2.0 * B * p.T * C * sizeof(float));
  if (lean) {
    const int64_t N = B * p.T;
    const float* rs = atf(ws, p.o_rowstat);
    const int32_t* ra = reinterpret_cast<const int32_t*>(rs + (size_t)N * cdiv(C, 256) * 2);
    MelJob job;
    if (pf.armed) job = MelJob{pf.src, pf.sample_lens, pf.aug, pf.B, pf.L, pf.normalize, pf.out_btf, pf.dtype, pf.frames_out, pf.pct_out, pf.ws, pf.ws_bytes};
    // (the next step's features ride in the grid of the compact lattice when its emissions fit one workgroup's LDS)
    LASR_TRY(ctc_loss_lean_job(at(ws, p.o_logits), lean_ldc(m), rs, ra, m->lean_tiles, targets, lens, tgt_lens, B, p.T, C, p.S_max, C - 1,
                               nll_out, argmax_out, at(ws, p.o_d1), grads + m->b_dec, nullptr, at(ws, p.o_lean), p.lean_bytes,
                               pf.armed ? &job : nullptr, stream));
  } else if (pf.armed) {   // this step's loss and the next step's features in one grid
    LASR_TRY(ctc_loss_mel_src(logp_out, targets, lens, tgt_lens, B, p.T, C, p.S_max, C - 1, nll_out, atf(ws, p.o_glogits), nullptr,
                              at(ws, p.o_ctc), p.ctc_bytes, pf.src, pf.sample_lens, pf.aug, pf.B, pf.L, pf.normalize, nullptr,
                              pf.out_btf, pf.dtype, pf.frames_out, pf.pct_out, pf.ws, pf.ws_bytes, stream));
  } else {
    LASR_TRY(lasr_ctc_loss(logp_out, targets, lens, tgt_lens, B, p.T, C, p.S_max, C - 1, nll_out, atf(ws, p.o_glogits), nullptr,
                           at(ws, p.o_ctc), p.ctc_bytes, stream));
  }
  // dense head in bf16 with a narrow vocabulary: the batch mean rides in the launch that opens the backward (head_tail, fused.h)
  if (!lean && m->cfg.dtype == LASR_BF16 && C <= 256) { m->tail_nll = nll_out; m->tail_loss = loss_out; return 0; }
  return lasr_scale_sum_f32(nll_out, B, 1.0f / (float)B, loss_out, stream);
}

extern "C" int lasr_model_loss_backward(lasr_model_t* m, const float* params, float* buffers, const void* feats, const float* pct,
                                        const int64_t* targets, const int32_t* tgt_lens, int64_t B, int64_t T_in, int64_t S_max,
                                        float* logp_out, float* loss_out, float* nll_out, int32_t* argmax_out, float* grads,
                                        void* ws, size_t ws_bytes, void* stream) {
  LASR_TRY(forward_and_loss(m, params, buffers, feats, pct, targets, tgt_lens, B, T_in, S_max, logp_out, loss_out, nll_out, argmax_out, grads,
                            ws, ws_bytes, stream, "lasr_model_loss_backward"));
  return backward_from_glogits(m, params, feats, B, T_in, grads, ws, stream);
}

extern "C" int64_t lasr_model_num_units(const lasr_model_t* m) { return m ? (int64_t)m->units.size() : -1; }

extern "C" int lasr_model_unit_info(const lasr_model_t* m, int64_t i, char* name, size_t name_cap) {
  LASR_CHECK_ARG(m && i >= 0 && i < (int64_t)m->units.size() && name && name_cap > 0, "lasr_model_unit_info: bad argument");
  strncpy(name, m->units[i].tap.c_str(), name_cap - 1);
  name[name_cap - 1] = 0;
  return 0;
}

extern "C" int lasr_model_loss_backward_partial(lasr_model_t* m, const float* params, float* buffers, const void* feats, const float* pct,
                                                const int64_t* targets, const int32_t* tgt_lens, int64_t B, int64_t T_in, int64_t S_max,
                                                float* logp_out, float* loss_out, float* nll_out, int32_t* argmax_out, float* grads,
                                                void* ws, size_t ws_bytes, int64_t unit_stop, void* stream) {
  LASR_CHECK_ARG(m && unit_stop >= 0 && unit_stop < (int64_t)m->units.size(), "lasr_model_loss_backward_partial: unit_stop");
  LASR_TRY(forward_and_loss(m, params, buffers, feats, pct, targets, tgt_lens, B, T_in, S_max, logp_out, loss_out, nll_out, argmax_out, grads,
                            ws, ws_bytes, stream, "lasr_model_loss_backward_partial"));
  return backward_from_glogits(m, params, feats, B, T_in, grads, ws, stream, true, -1, (int)unit_stop);
}

extern "C" int lasr_model_backward_continue(lasr_model_t* m, const float* params, const void* feats, int64_t B, int64_t T_in, float* grads,
                                            void* ws, size_t ws_bytes, int64_t unit_stop, void* stream) {
  LASR_CHECK_ARG(m && params && feats && grads && ws, "lasr_model_backward_continue: null pointer");
  LASR_CHECK_ARG(m->planned && m->plan.B == B && m->plan.T_in == T_in && m->bwd_next >= 0, "lasr_model_backward_continue: nothing pending");
  LASR_CHECK_ARG(unit_stop >= 0 && unit_stop <= m->bwd_next, "lasr_model_backward_continue: unit_stop");
  if (ws_bytes < m->plan.total) return fail(LASR_E_WORKSPACE, "lasr_model_backward_continue: workspace");
  return backward_from_glogits(m, params, feats, B, T_in, grads, ws, stream, false, m->bwd_next, (int)unit_stop);
}
