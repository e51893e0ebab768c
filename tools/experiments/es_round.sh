#!/bin/bash
# early-stop decode (the reference's semantics) on the resident engine: parity scatter by the variable lanes vs the round-start
# binary's gather syndrome; full GPU suite first
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/es2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for lib in default old; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  python - <<PY >> $O/es.jsonl 2>> $O/es.err
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
for w in ("basic", "rcq", "neural2d"):
    eng, dec, code = bench.build_decoder(w, dev)
    for snr in (2.0, 5.0, 6.5):
        llr = bench.make_llr(65536, code.n, snr, 1234, dev)
        for post in (False, True):
            run = lambda: eng.decode(llr, early_stop=True, want_bits=True, want_posterior=post)
            run(); torch.cuda.synchronize()
            ms = bench.event_ms(run, 5, torch)
            r = run()
            print(json.dumps({"lib": "$lib", "workload": w, "snr_db": snr, "posterior": post, "ms": round(ms, 4), "mean_iters": round(float(r.iterations.float().mean()), 3)}))
PY
done
unset LDPC_HIP_LIB
cat $O/es.jsonl
