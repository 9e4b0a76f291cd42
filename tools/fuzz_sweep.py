#!/usr/bin/env python3
"""A wider one-off run of tests/test_gpu_fuzz.py's differential sweep: other seeds, more configurations.
usage: python tools/fuzz_sweep.py <first seed> <number of seeds> [configs per seed]   (knobs via the environment)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    s0, ns = int(sys.argv[1]), int(sys.argv[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    import __graft_entry__ as ge
    import test_gpu_fuzz as tf
    pkg = ge.load_pkg()
    made = []

    def gpu():
        for g in made:      # one handle at a time: a sweep of a thousand corpora must not pile up device memory
            g.close()
        made.clear()
        made.append(pkg.GpuIndex(0))
        return made[-1]

    bad = skipped = 0
    t0 = time.time()
    for seed in range(s0, s0 + ns):
        for case in tf._configs(per, seed):
            try:
                tf.test_random_configuration_matches_oracle(gpu, case)
            except AssertionError as e:
                bad += 1
                print("FAIL seed %d: %s" % (seed, str(e)[:400]), flush=True)
            except ValueError as e:   # a corpus the generator cannot make (more queries than base vectors)
                skipped += 1
        print("seed %d done, %d failures so far, %.0fs" % (seed, bad, time.time() - t0), flush=True)
    print("fuzz sweep: %d configurations (%d skipped by the generator), %d failures" % (ns * per, skipped, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
