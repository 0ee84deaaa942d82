cd $GRAFT_REPO_ROOT
bash tools/pmc_step.sh gpurun_out/pmc_r03 > gpurun_out/pmc_r03.log 2>&1; python3 tools/pmc_step_summary.py gpurun_out/pmc_r03 gpurun_out/r03 >> gpurun_out/pmc_r03.log 2>&1
bash tools/pmc_xattn.sh gpurun_out/pmc_xattn > gpurun_out/r03_pmc_xattn.txt 2>&1
python bench.py --decode > gpurun_out/r03_decode.json 2> gpurun_out/r03_decode.err
python bench.py --fp8 --batch 512 --seq-len 256 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03_fp8_b512_l256.json 2> gpurun_out/r03_fp8_b512_l256.err
python bench.py --batch 512 --seq-len 256 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03_bf16_b512_l256.json 2> gpurun_out/r03_bf16_b512_l256.err
tail -3 gpurun_out/pmc_r03.log; tail -12 gpurun_out/r03_pmc_xattn.txt; head -c 600 gpurun_out/r03_decode.json; echo; head -c 400 gpurun_out/r03_fp8_b512_l256.json; tail -2 gpurun_out/r03_fp8_b512_l256.err; head -c 400 gpurun_out/r03_bf16_b512_l256.json; tail -2 gpurun_out/r03_bf16_b512_l256.err
