#!/bin/bash
# One GPU call that regenerates the round's committed evidence under gpurun_out/<tag>/ (copy what is to be judged to profiles/):
#   <tag>_train_step.md         rocprofv3 --kernel-trace --stats of the training leg of bench.py
#   <tag>_bigvgan.md            ... of tools/bench_vocoder.py
#   <tag>_hbm_traffic_*.json    FETCH_SIZE / WRITE_SIZE passes (separate, with --kernel-trace only) -> tools/pmc_traffic.py
#   <tag>_pmc_sq_mfma.txt       SQ wave-state / MFMA-busy counters of the training leg -> tools/pmc_summary.py
tag=${1:-r05}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TRAIN="python $R/bench.py --steps 20 --warmup 5 --preheat 0 --no-cpu-baseline --no-vocoder --no-app"
TRAIN2="python $R/bench.py --steps 2 --warmup 1 --preheat 0 --no-cpu-baseline --no-vocoder --no-app"
VOC="python $R/tools/bench_vocoder.py --iters 2"
rm -rf /tmp/p_*
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_train -o t -- $TRAIN > $out/train.log 2>&1
python $R/tools/prof_summary.py /tmp/p_train $out/${tag}_train_step.md "training leg of bench.py (20 timed + 5 warm-up + 1 instrumented step), round ${tag#r0}" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_voc -o v -- python $R/tools/bench_vocoder.py --iters 3 > $out/voc.log 2>&1
python $R/tools/prof_summary.py /tmp/p_voc $out/${tag}_bigvgan.md "tools/bench_vocoder.py --iters 3 (5 forwards of 64 x 1000 frames, bf16), round ${tag#r0}" > /dev/null 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_ft -- $TRAIN2 > $out/pmc_ft.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_wt -- $TRAIN2 > $out/pmc_wt.log 2>&1
python $R/tools/pmc_traffic.py /tmp/p_ft /tmp/p_wt $out/${tag}_hbm_traffic_train.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-app" 4 > $out/traffic_train.txt 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fv -- $VOC > $out/pmc_fv.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_wv -- $VOC > $out/pmc_wv.log 2>&1
python $R/tools/pmc_traffic.py /tmp/p_fv /tmp/p_wv $out/${tag}_hbm_traffic_bigvgan.json "python tools/bench_vocoder.py --iters 2" 4 > $out/traffic_voc.txt 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_sq -- $TRAIN2 > $out/pmc_sq.log 2>&1
(echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-app"; echo "# round ${tag#r0}; mean per dispatch (tools/pmc_summary.py); same counters as profiles/r02b_pmc_sq_mfma.txt"; python $R/tools/pmc_summary.py /tmp/p_sq | head -400) > $out/${tag}_pmc_sq_mfma.txt 2>&1
# the largest idle gaps of the main stream (tools/prof_gaps.py on the training trace) and the app path (config 5)
python $R/tools/prof_gaps.py /tmp/p_train > $out/${tag}_gaps.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_app -o a -- python $R/tools/bench_app_path.py > $out/app.log 2>&1
python $R/tools/prof_summary.py /tmp/p_app $out/${tag}_app_path.md "tools/bench_app_path.py (config 5: 32 prompts -> waveform, 2 warm-up + 3 timed batches), round ${tag#r0}" > /dev/null 2>&1
# BigVGAN SQ counters (VALU / MFMA busy of the fused layer kernels)
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_sqv -- $VOC > $out/pmc_sqv.log 2>&1
(echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -- python tools/bench_vocoder.py --iters 2"; echo "# round ${tag#r0}; mean per dispatch (tools/pmc_summary.py)"; python $R/tools/pmc_summary.py /tmp/p_sqv | head -120) > $out/${tag}_pmc_sq_bigvgan.txt 2>&1
ls -la $out
