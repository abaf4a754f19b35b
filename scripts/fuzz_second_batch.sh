# the three campaigns that found bugs in round 6 (call sequences, target routes, everything on one context) once more with new seeds: gpurun_out/big2/
# (profiles/r06_fuzz_second_batch.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/big2
timeout 330 python3 tests/fuzz/fuzz_api.py 2500 41 45 > gpurun_out/big2/api.json 2> gpurun_out/big2/api.err
timeout 150 python3 tests/fuzz/fuzz_modes.py 600 42 200000 > gpurun_out/big2/modes.json 2> gpurun_out/big2/modes.err
timeout 150 python3 tests/fuzz/fuzz_one_context.py 100 49 50 > gpurun_out/big2/one_context.json 2> gpurun_out/big2/one_context.err
for f in api modes one_context; do echo "$f: $(tail -c 400 gpurun_out/big2/$f.json | tr -d '\n' | tail -c 330)"; done
