"""Write a small synthetic corpus: 16-bit PCM 16 kHz wavs + JSON-lines manifests in the reference's
format ({"audio_filepath","duration","text"}, scripts/get_libri.py:135).  BASELINE config 1/2 data."""
import argparse
import json
import os
import wave

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="synth_data")
    ap.add_argument("--n-train", type=int, default=8)
    ap.add_argument("--n-dev", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--ragged", action="store_true", help="durations uniform in [2, seconds]")
    ap.add_argument("--labels", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "labels.txt"))
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    labels = [c.strip() for c in open(a.labels, encoding="utf-8").readlines()]
    rng = np.random.default_rng(a.seed)
    os.makedirs(a.out, exist_ok=True)
    for split, n in (("train", a.n_train), ("dev", a.n_dev)):
        with open(os.path.join(a.out, split + ".json"), "w", encoding="utf-8") as mf:
            for i in range(n):
                secs = float(rng.uniform(min(2.0, 0.5 * a.seconds), a.seconds)) if a.ragged else a.seconds
                L = int(secs * 16000)
                pcm = np.clip(0.1 * rng.standard_normal(L) * 32768, -32768, 32767).astype("<i2")
                path = os.path.abspath(os.path.join(a.out, "%s_%04d.wav" % (split, i)))
                with wave.open(path, "wb") as w:
                    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
                S = max(1, int(2.8 * secs))
                ids = rng.integers(0, len(labels), S)
                for s in range(1, S):              # no adjacent repeats: CTC-feasible for any T' >= S (note N9)
                    if ids[s] == ids[s - 1]:
                        ids[s] = (ids[s] + 1) % len(labels)
                mf.write(json.dumps({"audio_filepath": path, "duration": L / 16000.0, "text": "".join(labels[j] for j in ids)},
                                    ensure_ascii=False) + "\n")
    print("wrote", a.out)


if __name__ == "__main__":
    main()
