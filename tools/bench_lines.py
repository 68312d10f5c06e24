"""Prints the headline fields of the bench lines in the given log files (development aid)."""
import json
import sys
for f in sys.argv[1:]:
    try:
        for l in open(f):
            if l.startswith('{'):
                j = json.loads(l)
                r = j.get('roofline', {})
                two = (j.get('two_data_sets_in_flight') or {}).get('ms_per_step')
                print(f"{f.split('/')[-1]:28s} {j['config']['workload']:26s} ms/step {j['ms_per_step']:.4f}  kernel {r.get('kernel_ms') or 0:.4f}  frac {r.get('frac') or 0:.3f}  2sets {two or 0:.4f}")
    except OSError as e:
        print(f, e)
