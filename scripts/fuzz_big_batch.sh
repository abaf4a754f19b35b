# one batch of every differential campaign (tests/fuzz/*.py) with new seeds, on the GPU box: reports under gpurun_out/big/ (profiles/r06_fuzz_big_batch.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/big
timeout 900 python3 tests/fuzz/fuzz_api.py 2000 21 45 > gpurun_out/big/api.json 2> gpurun_out/big/api.err
timeout 600 python3 tests/fuzz/fuzz_modes.py 1500 22 200000 > gpurun_out/big/modes.json 2> gpurun_out/big/modes.err
timeout 600 python3 tests/fuzz/fuzz_oracle.py 3000 23 100000 > gpurun_out/big/oracle.json 2> gpurun_out/big/oracle.err
timeout 400 python3 tests/fuzz/fuzz_pre.py 1500 24 > gpurun_out/big/pre.json 2> gpurun_out/big/pre.err
timeout 400 python3 tests/fuzz/fuzz_dependent.py 400 25 300000 > gpurun_out/big/dependent.json 2> gpurun_out/big/dependent.err
timeout 400 python3 tests/fuzz/fuzz_map.py 600 26 40 > gpurun_out/big/map.json 2> gpurun_out/big/map.err
timeout 400 python3 tests/fuzz/fuzz_sequence.py 200 27 8 > gpurun_out/big/sequence.json 2> gpurun_out/big/sequence.err
timeout 300 python3 tests/fuzz/fuzz_next_rows.py 800 60 28 > gpurun_out/big/next_rows.json 2> gpurun_out/big/next_rows.err
timeout 300 python3 tests/fuzz/fuzz_one_context.py 150 29 50 > gpurun_out/big/one_context.json 2> gpurun_out/big/one_context.err
for f in api modes oracle pre dependent map sequence next_rows one_context; do echo "$f: $(tail -c 400 gpurun_out/big/$f.json | tr -d '\n' | tail -c 330)"; done
