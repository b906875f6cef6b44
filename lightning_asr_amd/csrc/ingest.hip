// Host-side ingest: the PCM of a batch of 16-bit wav files decoded by a few threads straight into ONE caller buffer (a pinned
// ring slot), so the batch goes to the GPU as a single int16 H2D copy and lasr_mel_fwd_src scales it on the device.
// The routines themselves live in host_io.h (plain C++, no HIP header: the same source is built as a g++ ASan / UBSan fuzz
// binary by tests/test_sanitize_cpu.py); this file is their C ABI.
#include "common.h"
#include "host_io.h"

static_assert(lasr::host::kLenLead == LASR_LEN_LEAD, "host_io.h and include/lasr.h disagree on the lead-in flag");

using namespace lasr;

static int host_rc(int rc, const std::string& err) {
  if (rc == host::kOk) return 0;
  return fail(rc == host::kErrWorkspace ? LASR_E_WORKSPACE : LASR_E_ARG, "%s", err.c_str());
}

extern "C" int lasr_wav_info(const char* path, int64_t* n_frames, int32_t* n_channels, int32_t* sample_rate, int32_t* bits) {
  std::string err;
  return host_rc(host::wav_info(path, n_frames, n_channels, sample_rate, bits, &err), err);
}

extern "C" int lasr_wav_read_batch(const char* const* paths, int64_t n, const double* crop_u, double crop_weight, int16_t* out,
                                   int64_t out_capacity, int64_t* ld_out, int32_t* lens_out, int32_t expect_rate, int n_threads,
                                   int lead_in) {
  std::string err;
  return host_rc(host::wav_read_batch(paths, n, crop_u, crop_weight, out, out_capacity, ld_out, lens_out, expect_rate, n_threads, lead_in, &err), err);
}
