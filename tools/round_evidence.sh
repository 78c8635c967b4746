#!/bin/bash
# One pass over the evidence a round keeps under profiles/: kernel-trace summaries of both workloads, hardware counters of
# the halo conv on two layer shapes, memory-side traffic per kernel, and the single-GPU rehearsals of the multi-GPU paths.
# Usage (GPU box, repo root): bash tools/round_evidence.sh <tag>      -> gpurun_out/<tag>/...
set -o pipefail
tag=${1:-evidence}
out=gpurun_out/$tag
mkdir -p $out
bash tools/prof_round.sh $tag || exit 1
bash tools/pmc_conv.sh "512 16 16 256 256" > $out/pmc_conv_256.txt 2>&1 || { tail -5 $out/pmc_conv_256.txt; exit 1; }
bash tools/pmc_conv.sh "512 32 32 64 64" > $out/pmc_conv_64.txt 2>&1 || { tail -5 $out/pmc_conv_64.txt; exit 1; }
bash tools/pmc_traffic.sh train ${tag}_traffic_train > /dev/null || exit 1
bash tools/pmc_traffic.sh sample ${tag}_traffic_sample > /dev/null || exit 1
cp gpurun_out/${tag}_traffic_train/traffic_train.txt gpurun_out/${tag}_traffic_sample/traffic_sample.txt $out/
# multi-GPU control flow on one GPU: one rank over RCCL, then two ranks over gloo on the same device
timeout -k 10 200 python bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --force-dist > $out/dist_rccl_1rank.json 2> $out/dist_rccl_1rank.err || { tail -5 $out/dist_rccl_1rank.err; exit 1; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --workload train --steps 3 --warmup 1 --no-cpu-baseline --backend gloo --same-device --batchnorm sync > $out/dist_gloo_2rank.json 2> $out/dist_gloo_2rank.err || { tail -5 $out/dist_gloo_2rank.err; exit 1; }
echo done
