#!/usr/bin/env python3
"""One-line digest of a bench.py output file: python tools/show_bench.py <file> ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        line = [ln for ln in open(f) if ln.startswith("{")][-1]
        d = json.loads(line)
        r, t = d["roofline"], d["trsm_panel"]
        print(f"{f}: {d['value']} {d['unit']} ({d['ms_per_step']} ms/step) | bulk {r['achieved']} TF frac {r['frac']} "
              f"x{r['launches']} {r['avg_launch_ms']} ms | trsm alone {t.get('achieved_TFlops')} TF {t.get('achieved_GBps')} GB/s "
              f"in-situ {t.get('in_situ_TFlops')} | potrf {d['potrf_tile']['avg_ms']} ms | lookahead {d.get('update_lookahead')} | "
              f"residual ok {d.get('residual', {}).get('ok')}")
    except Exception as e:  # noqa: BLE001
        print(f"{f}: cannot parse ({e!r})")
        try:
            print(open(f).read()[-600:])
        except OSError:
            pass
