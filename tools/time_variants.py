#!/usr/bin/env python3
"""Times the fp32 stream pass of the large-batch workload with each library variant given on the command
line (one child process per variant, GATK_PAIRHMM_LIB selects the library); checks each variant's
likelihoods against the first one's.  Prints one line per variant: min and median kernel ms of the passes."""
import os
import subprocess
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gatk_amd import synth
    from gatk_amd.pairhmm import HipLoglessPairHMM
    b = synth.config_c3(num_reads=4096)
    hmm = HipLoglessPairHMM(0)
    hmm.set_option("timing", 1)
    res = hmm.upload([b])
    ms = []
    for _ in range(int(sys.argv[2])):
        res.compute(sync=True)
        ms.append(hmm.stats()["fp32_kernel_ms"])
    out = res.download()[0]
    ref_path = sys.argv[3]
    if os.path.exists(ref_path):
        ref = np.load(ref_path)
        same = bool(np.array_equal(ref, out))
    else:
        np.save(ref_path, out)
        same = True
    ms = sorted(ms[2:])
    print(f"min {ms[0]:.4f} median {ms[len(ms) // 2]:.4f} ms  identical_to_first={same}", flush=True)
    res.free()
    hmm.close()
    sys.exit(0)

ref = "/tmp/variant_ref.npy"
if os.path.exists(ref):
    os.remove(ref)
for rounds in range(2):  # every variant twice, interleaved, so that drift of the device clock shows
    for lib in sys.argv[1:]:
        env = dict(os.environ, GATK_PAIRHMM_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "14", ref], env=env, capture_output=True, text=True, timeout=120)
        print(f"{os.path.basename(lib):24s} {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
