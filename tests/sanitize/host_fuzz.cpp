// CPU sanitizer pass over the host routines that read untrusted files: built by tests/test_sanitize_cpu.py as
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -pthread tests/sanitize/host_fuzz.cpp
// against lightning_asr_amd/csrc/host_io.h - the SAME source liblasr.so compiles (ingest.hip / capi.hip wrap it).  No GPU, no HIP.
// Exit code 0 = every case behaved (a clean error or a correct read); any sanitizer report aborts with a non-zero code.
//
//   1. header variants the reader must accept (plain PCM, LIST chunk before fmt/data, odd-sized chunk with its pad byte,
//      WAVE_FORMAT_EXTENSIBLE, stereo, a streamed file whose data size is 0xFFFFFFFF) - samples compared with what was written;
//   2. a fuzz corpus: every truncation of a valid file's first 64 bytes, size fields forced to 0 / 1 / 0x7FFFFFFF / 0xFFFFFFFF,
//      zero and huge channel counts, 8 / 24 / 32 bits, non-PCM format tags, and 4 000 LCG-driven byte mutations - each through
//      wav_info and wav_read_batch (with and without the training-time crop + lead-in, 1 and 3 threads, tight buffers);
//   3. crop_slice with hostile uniforms (negative, > 1, NaN, infinities) stays inside the file;
//   4. the host Levenshtein distance: known answers, symmetry, identity, empty sides, bad arguments.
#include "../../lightning_asr_amd/csrc/host_io.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

using namespace lasr::host;

static int g_fail = 0;
#define CHECK(cond, ...)                                                         \
  do {                                                                           \
    if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } \
  } while (0)

typedef std::vector<unsigned char> Bytes;

static void put16(Bytes& b, uint16_t v) { b.push_back(v & 255); b.push_back(v >> 8); }
static void put32(Bytes& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((v >> (8 * i)) & 255); }
static void tag(Bytes& b, const char* t) { b.insert(b.end(), t, t + 4); }

struct Spec {
  int channels = 1, bits = 16, rate = 16000, fmt = 1;
  bool extensible = false, list_first = false, odd_chunk = false, bogus_data_size = false;
};

static Bytes make_wav(const std::vector<int16_t>& ch0, const Spec& s) {
  Bytes b;
  tag(b, "RIFF"); put32(b, 0); tag(b, "WAVE");
  if (s.list_first) { tag(b, "LIST"); put32(b, 10); tag(b, "INFO"); for (int i = 0; i < 6; ++i) b.push_back('x'); }
  tag(b, "fmt ");
  put32(b, s.extensible ? 40 : 16);
  put16(b, s.extensible ? 0xFFFE : (uint16_t)s.fmt); put16(b, (uint16_t)s.channels); put32(b, (uint32_t)s.rate);
  put32(b, (uint32_t)(s.rate * s.channels * s.bits / 8)); put16(b, (uint16_t)(s.channels * s.bits / 8)); put16(b, (uint16_t)s.bits);
  if (s.extensible) { put16(b, 22); put16(b, (uint16_t)s.bits); put32(b, 0); put16(b, (uint16_t)s.fmt); for (int i = 0; i < 14; ++i) b.push_back(0); }
  if (s.odd_chunk) { tag(b, "junk"); put32(b, 3); b.push_back(1); b.push_back(2); b.push_back(3); b.push_back(0); }   // 3 bytes + pad
  tag(b, "data");
  put32(b, s.bogus_data_size ? 0xFFFFFFFFu : (uint32_t)(ch0.size() * s.channels * 2));
  for (size_t i = 0; i < ch0.size(); ++i)
    for (int c = 0; c < s.channels; ++c) put16(b, (uint16_t)(c == 0 ? ch0[i] : (int16_t)(ch0[i] ^ (0x1111 * c))));
  const uint32_t riff = (uint32_t)b.size() - 8;
  for (int i = 0; i < 4; ++i) b[4 + i] = (riff >> (8 * i)) & 255;
  return b;
}

static std::string g_dir;
static std::string write_file(const char* name, const Bytes& b) {
  const std::string p = g_dir + "/" + name;
  FILE* f = fopen(p.c_str(), "wb");
  if (!f) { perror(p.c_str()); exit(2); }
  if (!b.empty()) fwrite(b.data(), 1, b.size(), f);
  fclose(f);
  return p;
}

// read ONE file through the batch reader into a guarded buffer of exactly `cap` samples (ASan watches both ends)
static int read_one(const std::string& path, const double* crop, int lead, int threads, int64_t cap, std::vector<int16_t>* row, int32_t* len_word,
                    std::string* err) {
  std::vector<int16_t> out((size_t)cap, (int16_t)0x5A5A);
  const char* paths[1] = {path.c_str()};
  int64_t ld = 0;
  const int rc = wav_read_batch(paths, 1, crop, 0.98, out.data(), cap, &ld, len_word, 0, threads, lead, err);
  if (rc == kOk) {
    CHECK(ld >= 2 && ld % 8 == 0 && ld <= cap, "ld=%lld cap=%lld", (long long)ld, (long long)cap);
    row->assign(out.begin(), out.begin() + ld);
  }
  return rc;
}

static void accepted_variants() {
  std::vector<int16_t> x(1000);
  uint32_t s = 12345;
  for (auto& v : x) { s = s * 1664525u + 1013904223u; v = (int16_t)(s >> 16); }
  const char* names[] = {"plain", "list_first", "odd_chunk", "extensible", "stereo", "bogus_size", "all"};
  for (int k = 0; k < 7; ++k) {
    Spec sp;
    sp.list_first = k == 1 || k == 6; sp.odd_chunk = k == 2 || k == 6; sp.extensible = k == 3 || k == 6;
    sp.channels = (k == 4 || k == 6) ? 2 : 1; sp.bogus_data_size = k == 5;
    const std::string p = write_file((std::string(names[k]) + ".wav").c_str(), make_wav(x, sp));
    int64_t nf = 0; int32_t ch = 0, rate = 0, bits = 0; std::string err;
    CHECK(wav_info(p.c_str(), &nf, &ch, &rate, &bits, &err) == kOk, "%s: %s", names[k], err.c_str());
    CHECK(nf == 1000 && ch == sp.channels && rate == 16000 && bits == 16, "%s: nf=%lld ch=%d", names[k], (long long)nf, ch);
    for (int threads = 1; threads <= 3; threads += 2) {
      std::vector<int16_t> row; int32_t lw = 0;
      CHECK(read_one(p, nullptr, 0, threads, 1000, &row, &lw, &err) == kOk, "%s: %s", names[k], err.c_str());
      CHECK(lw == 1000 && row.size() == 1000 && std::equal(x.begin(), x.end(), row.begin()), "%s: samples differ", names[k]);
      // the training-time crop with its lead-in sample: row = [x[loc-1], x[loc .. target)]
      const double u[2] = {0.5, 0.75};
      CHECK(read_one(p, u, 1, threads, 1008, &row, &lw, &err) == kOk, "%s crop: %s", names[k], err.c_str());
      const int64_t target = (int64_t)(1000 * (0.98 + 0.02 * 0.5)), loc = (int64_t)(0.75 * (1000 - target));
      CHECK((lw & (kLenLead - 1)) == target - loc && (lw & kLenLead) != 0, "%s crop: len word %x", names[k], lw);
      CHECK(std::equal(x.begin() + loc - 1, x.begin() + target, row.begin()), "%s crop: samples differ", names[k]);
      for (size_t i = (size_t)(target - loc + 1); i < row.size(); ++i) CHECK(row[i] == 0, "%s crop: tail not zeroed", names[k]);
    }
    // a buffer one row too small is refused, not overrun
    std::vector<int16_t> row; int32_t lw = 0;
    CHECK(read_one(p, nullptr, 0, 1, 999, &row, &lw, &err) == kErrWorkspace, "%s: tight buffer accepted", names[k]);
  }
  // a batch of three files of different lengths, 3 threads
  std::vector<std::string> ps;
  for (int n : {17, 1000, 333}) {
    std::vector<int16_t> y(x.begin(), x.begin() + n);
    ps.push_back(write_file(("batch" + std::to_string(n) + ".wav").c_str(), make_wav(y, Spec())));
  }
  const char* paths[3] = {ps[0].c_str(), ps[1].c_str(), ps[2].c_str()};
  std::vector<int16_t> out(3 * 1000, 0x5A5A);
  int64_t ld = 0; int32_t lens[3]; std::string err;
  CHECK(wav_read_batch(paths, 3, nullptr, 0.98, out.data(), 3000, &ld, lens, 16000, 3, 0, &err) == kOk, "batch: %s", err.c_str());
  CHECK(ld == 1000 && lens[0] == 17 && lens[1] == 1000 && lens[2] == 333, "batch: ld=%lld", (long long)ld);
  CHECK(out[16] == x[16] && out[17] == 0 && out[1000 + 999] == x[999] && out[2000 + 332] == x[332] && out[2000 + 333] == 0, "batch rows");
  CHECK(wav_read_batch(paths, 3, nullptr, 0.98, out.data(), 3000, &ld, lens, 8000, 3, 0, &err) == kErrArg, "wrong rate accepted");
  CHECK(wav_read_batch(paths, 3, nullptr, 0.98, out.data(), 2999, &ld, lens, 0, 3, 0, &err) == kErrWorkspace, "tight batch buffer accepted");
}

// one hostile file through every entry point; whatever comes back must be a clean verdict
static void hammer(const std::string& p, uint32_t salt) {
  int64_t nf = -1; int32_t ch = -1, rate = -1, bits = -1; std::string err;
  const int rc = wav_info(p.c_str(), &nf, &ch, &rate, &bits, &err);
  if (rc == kOk) CHECK(nf >= 0 && ch >= 1 && ch <= kMaxChannels && bits == 16, "accepted: nf=%lld ch=%d bits=%d", (long long)nf, ch, bits);
  else CHECK(!err.empty(), "error without a message");
  const double crops[3][2] = {{0.0, 0.0}, {0.999, 0.999}, {(salt % 1000) / 1000.0, ((salt / 1000) % 1000) / 1000.0}};
  for (int k = 0; k < 4; ++k) {
    std::vector<int16_t> row; int32_t lw = 0;
    const int64_t cap = (k == 3) ? 8 : 4096;                  // k == 3: a buffer that only fits 8 samples
    const int r2 = read_one(p, k < 3 ? crops[k] : nullptr, k & 1, 1 + (k % 3), cap, &row, &lw, &err);
    if (r2 == kOk) {
      const int64_t n = lw & (kLenLead - 1), lead = (lw & kLenLead) ? 1 : 0;
      CHECK(n + lead <= (int64_t)row.size(), "len word %lld + %lld beyond the row (%zu)", (long long)n, (long long)lead, row.size());
      for (size_t i = (size_t)(n + lead); i < row.size(); ++i) CHECK(row[i] == 0, "tail not zeroed");
    }
  }
}

static void fuzz_corpus() {
  std::vector<int16_t> x(300);
  for (size_t i = 0; i < x.size(); ++i) x[i] = (int16_t)(i * 37 - 5000);
  const Bytes good = make_wav(x, Spec());
  int n_files = 0;
  // truncations of the header region and around the end of the data
  for (size_t cut = 0; cut <= 64; ++cut) hammer(write_file("fz.wav", Bytes(good.begin(), good.begin() + std::min(cut, good.size()))), (uint32_t)cut), ++n_files;
  for (size_t cut = good.size() - 5; cut < good.size(); ++cut) hammer(write_file("fz.wav", Bytes(good.begin(), good.begin() + cut)), (uint32_t)cut), ++n_files;
  // size fields: RIFF size (4), fmt size (16), data size (40)
  const uint32_t sizes[] = {0u, 1u, 2u, 15u, 16u, 17u, 39u, 41u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu};
  for (size_t at : {4u, 16u, 40u})
    for (uint32_t v : sizes) {
      Bytes b = good;
      for (int i = 0; i < 4; ++i) b[at + i] = (v >> (8 * i)) & 255;
      hammer(write_file("fz.wav", b), v ^ (uint32_t)at), ++n_files;
    }
  // fmt fields: format tag (20), channels (22), rate (24), bits (34)
  for (uint16_t v : {0, 2, 3, 6, 7, 0xFFFE, 0xFFFF}) { Bytes b = good; b[20] = v & 255; b[21] = v >> 8; hammer(write_file("fz.wav", b), v), ++n_files; }
  for (uint16_t v : {0, 2, 3, 255, 256, 257, 4096, 0x7FFF, 0xFFFF}) { Bytes b = good; b[22] = v & 255; b[23] = v >> 8; hammer(write_file("fz.wav", b), v), ++n_files; }
  for (uint32_t v : {0u, 1u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu}) { Bytes b = good; for (int i = 0; i < 4; ++i) b[24 + i] = (v >> (8 * i)) & 255; hammer(write_file("fz.wav", b), v), ++n_files; }
  for (uint16_t v : {0, 8, 12, 24, 32, 64, 0xFFFF}) { Bytes b = good; b[34] = v & 255; b[35] = v >> 8; hammer(write_file("fz.wav", b), v), ++n_files; }
  // data before fmt, two fmt chunks, no data chunk, an extensible header cut short
  { Bytes b(good.begin(), good.begin() + 12); b.insert(b.end(), good.begin() + 36, good.end()); b.insert(b.end(), good.begin() + 12, good.begin() + 36); hammer(write_file("fz.wav", b), 1), ++n_files; }
  { Bytes b(good.begin(), good.begin() + 36); b.insert(b.end(), good.begin() + 12, good.end()); hammer(write_file("fz.wav", b), 2), ++n_files; }
  { Bytes b(good.begin(), good.begin() + 36); hammer(write_file("fz.wav", b), 3), ++n_files; }
  { Spec sp; sp.extensible = true; Bytes b = make_wav(x, sp); b[16] = 24; hammer(write_file("fz.wav", b), 4), ++n_files; }
  // an empty file, a directory, a path that does not exist
  hammer(write_file("fz.wav", Bytes()), 5), ++n_files;
  hammer(g_dir, 6);
  hammer(g_dir + "/does-not-exist.wav", 7);
  // LCG-driven mutations: 1-4 bytes of the first 64 replaced, sometimes the tail cut as well
  uint32_t s = 20261005u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int it = 0; it < 4000; ++it) {
    Spec sp; sp.channels = 1 + rnd() % 3; sp.extensible = rnd() % 4 == 0; sp.list_first = rnd() % 4 == 0; sp.odd_chunk = rnd() % 4 == 0;
    Bytes b = make_wav(x, sp);
    const int nm = 1 + rnd() % 4;
    for (int m = 0; m < nm; ++m) {
      const size_t at = rnd() % std::min<size_t>(b.size(), 96);
      const uint32_t how = rnd() % 4;
      b[at] = how == 0 ? 0 : how == 1 ? 0xFF : how == 2 ? (unsigned char)(b[at] ^ (1u << (rnd() % 8))) : (unsigned char)rnd();
    }
    if (rnd() % 5 == 0) b.resize(rnd() % (b.size() + 1));
    hammer(write_file("fz.wav", b), rnd()), ++n_files;
  }
  fprintf(stderr, "fuzz corpus: %d files\n", n_files);
}

static void crop_slices() {
  const double inf = std::numeric_limits<double>::infinity(), nan = std::nan("");
  const double vals[] = {0.0, 0.5, 0.999999, 1.0, 1.5, -0.5, 1e300, -1e300, inf, -inf, nan};
  for (int64_t length : {0ll, 1ll, 2ll, 160000ll, (1ll << 40)})
    for (double u0 : vals) for (double u1 : vals) for (double w : {0.98, 0.1, 0.0, 1.0, -3.0, 7.0, nan}) {
      const double u[2] = {u0, u1};
      int64_t first = -1, count = -1;
      crop_slice(length, u, w, &first, &count);
      CHECK(first >= 0 && count >= 0 && first + count <= length, "crop_slice(%lld, %g, %g, %g) -> %lld + %lld", (long long)length, u0, u1, w,
            (long long)first, (long long)count);
    }
  // and the reference's arithmetic on ordinary draws (data_module.py:138-148)
  const double u[2] = {0.25, 0.5};
  int64_t first = 0, count = 0;
  crop_slice(160000, u, 0.98, &first, &count);
  const int64_t target = (int64_t)(160000 * (0.98 + 0.02 * 0.25)), loc = (int64_t)(0.5 * (160000 - target));
  CHECK(first == loc && count == target - loc, "crop_slice disagrees with sub_secquence");
}

static void levenshtein() {
  auto d = [](std::vector<int32_t> a, std::vector<int32_t> b) { return edit_distance(a.empty() ? nullptr : a.data(), (int64_t)a.size(), b.empty() ? nullptr : b.data(), (int64_t)b.size()); };
  auto str = [](const char* s) { std::vector<int32_t> v; for (; *s; ++s) v.push_back(*s); return v; };
  CHECK(d(str("kitten"), str("sitting")) == 3, "kitten/sitting");
  CHECK(d(str("flaw"), str("lawn")) == 2, "flaw/lawn");
  CHECK(d(str("intention"), str("execution")) == 5, "intention/execution");
  CHECK(d({}, {}) == 0 && d(str("abc"), {}) == 3 && d({}, str("abcd")) == 4 && d(str("abc"), str("abc")) == 0, "empty / identity");
  int32_t one = 1;
  CHECK(edit_distance(nullptr, 3, &one, 1) == -1 && edit_distance(&one, -1, &one, 1) == -1 && edit_distance(&one, 1, nullptr, 2) == -1, "bad arguments");
  uint32_t s = 99;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 10; };
  for (int it = 0; it < 300; ++it) {
    std::vector<int32_t> a(rnd() % 40), b(rnd() % 40), c(rnd() % 40);
    for (auto& v : a) v = rnd() % 5; for (auto& v : b) v = rnd() % 5; for (auto& v : c) v = rnd() % 5;
    const int64_t ab = d(a, b), ba = d(b, a), ac = d(a, c), cb = d(c, b);
    const int64_t diff = std::llabs((long long)a.size() - (long long)b.size());
    CHECK(ab == ba && ab >= diff && ab <= (int64_t)std::max(a.size(), b.size()) && ab <= ac + cb, "metric properties");
  }
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: host_fuzz <scratch dir>\n"); return 2; }
  g_dir = argv[1];
  accepted_variants();
  fuzz_corpus();
  crop_slices();
  levenshtein();
  if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
  printf("host_fuzz ok\n");
  return 0;
}
