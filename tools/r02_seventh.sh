set -o pipefail
mkdir -p gpurun_out/r02g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r02g/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --rehearse --streams 8 --settle 40 --sustain 20 > gpurun_out/r02g/rehearse2.json 2> gpurun_out/r02g/rehearse2.err; echo "rehearse rc=$?"; tail -c 1500 gpurun_out/r02g/rehearse2.json; tail -5 gpurun_out/r02g/rehearse2.err
