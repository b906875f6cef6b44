"""CPU sanitizer pass (ASan + UBSan; GPU sanitizers are not available on the pool) over the host-side routines that parse untrusted
files - the RIFF/WAVE walk, the training-time crop slice and the batch reader behind `lasr_wav_info` / `lasr_wav_read_batch`
(the reference's `torchaudio.load` + `sub_secquence`, /root/reference/data_module.py:138-159) - and the host Levenshtein distance
behind `lasr_edit_distance` (utils/asr_metrics.py:54,220).  The routines live in lightning_asr_amd/csrc/host_io.h, the SAME source
liblasr.so compiles; tests/sanitize/host_fuzz.cpp drives them with the accepted header variants, a fuzz corpus of > 4 000 hostile
files (truncations, size fields of 0 / 0xFFFFFFFF, zero / huge channel counts, odd sizes, byte mutations), hostile crop uniforms
and known-answer edit distances."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined,float-cast-overflow,float-divide-by-zero", "-fno-sanitize-recover=all",
         "-pthread"]


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_file_readers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_fuzz")
    b = subprocess.run(["g++"] + FLAGS + [os.path.join(ROOT, "tests", "sanitize", "host_fuzz.cpp"), "-o", exe], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-3000:]
    scratch = tmp_path / "corpus"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, str(scratch)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "host_fuzz ok" in r.stdout and "fuzz corpus:" in r.stderr


def test_library_wraps_the_sanitized_source():
    """ingest.hip / capi.hip hold no parsing code of their own: they include host_io.h and wrap it"""
    ing = open(os.path.join(ROOT, "lightning_asr_amd", "csrc", "ingest.hip")).read()
    cap = open(os.path.join(ROOT, "lightning_asr_amd", "csrc", "capi.hip")).read()
    assert '#include "host_io.h"' in ing and "host::wav_read_batch(" in ing and "pread(" not in ing
    assert '#include "host_io.h"' in cap and "host::edit_distance(" in cap
    hdr = open(os.path.join(ROOT, "lightning_asr_amd", "csrc", "host_io.h")).read()
    assert "hip" not in hdr.split("#pragma once", 1)[1].lower().replace("ingest.hip", "").replace("capi.hip", "")
