// Host-side routines that read UNTRUSTED files (the RIFF/WAVE walk, the training-time crop slice, the batch reader) and the host
// Levenshtein distance - plain C++17, no HIP header, so that the same source also builds as a g++ -fsanitize=address,undefined
// test binary (tests/sanitize/host_fuzz.cpp, run by tests/test_sanitize_cpu.py; GPU sanitizers are not available on the pool).
// ingest.hip / capi.hip wrap these behind the C ABI (lasr_wav_info, lasr_wav_read_batch, lasr_edit_distance).
// Replaces `torchaudio.load` in the DataLoader workers of the reference (data_module.py:153, conf/conf.yaml:14 num_worker: 6)
// and the training-time random sub-sequence (data_module.py:138-148,158-159), which is a slice of the file.  No device work.
#pragma once
#include <algorithm>
#include <atomic>
#include <stdint.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace lasr {
namespace host {

constexpr int32_t kLenLead = 1 << 30;      // == LASR_LEN_LEAD (include/lasr.h; static_assert in ingest.hip)
constexpr int kMaxChannels = 256;          // a header that claims more is refused (the de-interleave buffer is 8 KB per channel)
enum { kOk = 0, kErrArg = 1, kErrWorkspace = 2 };      // mapped to LASR_E_* by the wrappers

struct WavInfo {
  int64_t data_off = 0;     // byte offset of the first sample frame
  int64_t n_frames = 0;
  int32_t channels = 0, rate = 0, bits = 0;
};

static inline uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// RIFF/WAVE header walk: "fmt " (PCM = 1, or WAVE_FORMAT_EXTENSIBLE with a PCM sub-format) and "data".  Returns "" or a message.
// Every size field is read as unsigned 32-bit into int64 arithmetic and checked against the file's real size.
static inline std::string wav_open(const char* path, int* fd_out, WavInfo* w) {
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return std::string("cannot open ") + path;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return std::string("cannot stat ") + path; }
  unsigned char h[12];
  if (pread(fd, h, 12, 0) != 12 || memcmp(h, "RIFF", 4) != 0 || memcmp(h + 8, "WAVE", 4) != 0) { close(fd); return std::string("not a RIFF/WAVE file: ") + path; }
  int64_t off = 12;
  bool have_fmt = false;
  while (off + 8 <= (int64_t)st.st_size) {
    unsigned char ch[8];
    if (pread(fd, ch, 8, off) != 8) break;
    const int64_t sz = rd32(ch + 4);
    if (memcmp(ch, "fmt ", 4) == 0) {
      unsigned char f[40];
      const int64_t n = sz < 40 ? sz : 40;
      if (n < 16 || pread(fd, f, (size_t)n, off + 8) != n) { close(fd); return std::string("short fmt chunk: ") + path; }
      int fmt = rd16(f);
      w->channels = rd16(f + 2); w->rate = (int32_t)(rd32(f + 4) & 0x7FFFFFFFu); w->bits = rd16(f + 14);
      if (fmt == 0xFFFE && n >= 26) fmt = rd16(f + 24);       // extensible: first two bytes of the sub-format GUID
      if (fmt != 1) { close(fd); return std::string("only integer PCM wav is supported: ") + path; }
      have_fmt = true;
    } else if (memcmp(ch, "data", 4) == 0) {
      if (!have_fmt) { close(fd); return std::string("data chunk before fmt chunk: ") + path; }
      if (w->bits != 16 || w->channels < 1) { close(fd); return std::string("only 16-bit PCM wav is supported: ") + path; }
      if (w->channels > kMaxChannels) { close(fd); return std::string("more channels than a wav file has: ") + path; }
      int64_t bytes = sz;
      if (off + 8 + bytes > (int64_t)st.st_size) bytes = (int64_t)st.st_size - off - 8;   // streamed files carry a bogus size
      w->data_off = off + 8;
      w->n_frames = bytes / (2 * w->channels);
      *fd_out = fd;
      return "";
    }
    off += 8 + sz + (sz & 1);
  }
  close(fd);
  return std::string("no data chunk: ") + path;
}

struct Job {
  int fd = -1;
  WavInfo w;
  int64_t first = 0, count = 0;    // slice of the file this batch takes
  int lead = 0;                    // 1: the row starts with file sample first - 1 (LASR_LEN_LEAD)
  std::string err;
};

// data_module.py:138-148 sub_secquence(x, weight): target_length = int(length * U(weight, 1)); location = int(U(0, length -
// target_length)); x[:, location:target_length] - the slice END is target_length (reference behaviour, kept).  u = the two uniforms;
// values outside [0, 1] (a caller's bug, NaN included) are clamped so that the slice always lies inside the file.
static inline void crop_slice(int64_t length, const double* u, double weight, int64_t* first, int64_t* count) {
  const double u0 = u[0] >= 0.0 ? (u[0] <= 1.0 ? u[0] : 1.0) : 0.0, u1 = u[1] >= 0.0 ? (u[1] <= 1.0 ? u[1] : 1.0) : 0.0;
  const double wgt = weight >= 0.0 ? (weight <= 1.0 ? weight : 1.0) : 0.0;
  int64_t target = (int64_t)((double)length * (wgt + (1.0 - wgt) * u0));
  target = std::min(std::max<int64_t>(target, 0), length);
  int64_t loc = (int64_t)(u1 * (double)(length - target));
  loc = std::min(std::max<int64_t>(loc, 0), length);
  *first = loc;
  *count = target > loc ? target - loc : 0;
}

template <typename F>
static inline void parallel_for(int64_t n, int n_threads, F&& body) {
  if (n_threads <= 1 || n <= 1) { for (int64_t i = 0; i < n; ++i) body(i); return; }
  std::atomic<int64_t> next(0);
  auto run = [&]() { for (int64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) body(i); };
  std::vector<std::thread> th;
  const int nt = (int)std::min<int64_t>(n_threads, n);
  for (int t = 1; t < nt; ++t) th.emplace_back(run);
  run();
  for (auto& t : th) t.join();
}

static inline int wav_info(const char* path, int64_t* n_frames, int32_t* n_channels, int32_t* sample_rate, int32_t* bits, std::string* err) {
  if (!path) { *err = "lasr_wav_info: null path"; return kErrArg; }
  int fd = -1;
  WavInfo w;
  const std::string e = wav_open(path, &fd, &w);
  if (!e.empty()) { *err = "lasr_wav_info: " + e; return kErrArg; }
  close(fd);
  if (n_frames) *n_frames = w.n_frames;
  if (n_channels) *n_channels = w.channels;
  if (sample_rate) *sample_rate = w.rate;
  if (bits) *bits = w.bits;
  return kOk;
}

static inline int wav_read_batch(const char* const* paths, int64_t n, const double* crop_u, double crop_weight, int16_t* out,
                                 int64_t out_capacity, int64_t* ld_out, int32_t* lens_out, int32_t expect_rate, int n_threads,
                                 int lead_in, std::string* err) {
  if (!(paths && n > 0 && out && ld_out && lens_out && out_capacity > 0)) { *err = "lasr_wav_read_batch: bad argument"; return kErrArg; }
  std::vector<Job> jobs((size_t)n);
  // pass 1: headers -> the slice every file contributes
  parallel_for(n, n_threads, [&](int64_t i) {
    Job& j = jobs[(size_t)i];
    if (!paths[i]) { j.err = "null path"; return; }
    j.err = wav_open(paths[i], &j.fd, &j.w);
    if (!j.err.empty()) return;
    if (expect_rate > 0 && j.w.rate != expect_rate) { j.err = std::string("unexpected sample rate in ") + paths[i]; return; }
    j.first = 0; j.count = j.w.n_frames;
    if (crop_u) crop_slice(j.w.n_frames, crop_u + 2 * i, crop_weight, &j.first, &j.count);
    j.lead = (lead_in && j.first > 0 && j.count > 0) ? 1 : 0;
    if (j.count + j.lead >= kLenLead) j.err = std::string("too long: ") + paths[i];
  });
  int64_t lmax = 0;
  const Job* bad = nullptr;
  for (const Job& j : jobs) { if (!j.err.empty() && !bad) bad = &j; if (j.err.empty()) lmax = std::max(lmax, j.count + j.lead); }
  const int64_t ld = (std::max<int64_t>(lmax, 2) + 7) / 8 * 8;      // rows stay 16-byte aligned
  if (!bad && (ld > out_capacity || n > out_capacity / ld)) {
    for (Job& j : jobs) if (j.fd >= 0) close(j.fd);
    *err = "lasr_wav_read_batch: " + std::to_string((long long)n) + " x " + std::to_string((long long)ld) + " samples do not fit the buffer (" +
           std::to_string((long long)out_capacity) + ")";
    return kErrWorkspace;
  }
  if (bad) {
    *err = "lasr_wav_read_batch: " + bad->err;
    for (Job& j : jobs) if (j.fd >= 0) close(j.fd);
    return kErrArg;
  }
  // pass 2: PCM of channel 0 -> out[i][0 .. count), zeros up to ld
  parallel_for(n, n_threads, [&](int64_t i) {
    Job& j = jobs[(size_t)i];
    int16_t* row = out + i * ld;
    const int ch = j.w.channels;
    int64_t got = 0;
    const int64_t first = j.first - j.lead, count = j.count + j.lead;    // the row: [lead-in sample] + the slice
    if (ch == 1) {
      const int64_t want = count * 2;
      int64_t done = 0;
      while (done < want) {
        const ssize_t r = pread(j.fd, reinterpret_cast<char*>(row) + done, (size_t)(want - done), j.w.data_off + first * 2 + done);
        if (r <= 0) break;
        done += r;
      }
      got = done / 2;
    } else {   // interleaved: keep channel 0 (the reference feeds row 0 of torchaudio.load's (channels, L) result)
      std::vector<int16_t> tmp((size_t)4096 * ch);
      while (got < count) {
        const int64_t fr = std::min<int64_t>(4096, count - got);
        const ssize_t r = pread(j.fd, tmp.data(), (size_t)(fr * ch * 2), j.w.data_off + (first + got) * ch * 2);
        if (r < (ssize_t)(fr * ch * 2)) break;
        for (int64_t t = 0; t < fr; ++t) row[got + t] = tmp[(size_t)(t * ch)];
        got += fr;
      }
    }
    if (got < count) j.err = std::string("short read: ") + paths[i];
    memset(row + got, 0, (size_t)(ld - got) * sizeof(int16_t));
    lens_out[i] = (int32_t)j.count | (j.lead ? kLenLead : 0);
    close(j.fd);
    j.fd = -1;
  });
  for (const Job& j : jobs)
    if (!j.err.empty()) { *err = "lasr_wav_read_batch: " + j.err; return kErrArg; }
  *ld_out = ld;
  return kOk;
}

// ---- host-side Levenshtein distance on token-id sequences (WER/CER: utils/asr_metrics.py:54,220 call editdistance.eval on word /
// character lists; the host maps words to ids first).  -1: bad argument.
static inline int64_t edit_distance(const int32_t* a, int64_t na, const int32_t* b, int64_t nb) {
  if (na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b)) return -1;
  std::vector<int64_t> prev((size_t)nb + 1), cur((size_t)nb + 1);
  for (int64_t j = 0; j <= nb; ++j) prev[(size_t)j] = j;
  for (int64_t i = 1; i <= na; ++i) {
    cur[0] = i;
    for (int64_t j = 1; j <= nb; ++j) {
      const int64_t sub = prev[(size_t)j - 1] + (a[i - 1] != b[j - 1]);
      const int64_t del = prev[(size_t)j] + 1, ins = cur[(size_t)j - 1] + 1;
      cur[(size_t)j] = sub < del ? (sub < ins ? sub : ins) : (del < ins ? del : ins);
    }
    prev.swap(cur);
  }
  return prev[(size_t)nb];
}

}  // namespace host
}  // namespace lasr
