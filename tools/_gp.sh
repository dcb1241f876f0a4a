set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_asm.py tests/test_gpu_ont.py -m gpu -x -q 2>&1 | tail -3 && python bench.py --lanes 1 --cpu-sample 0 --holdout 0 > gpurun_out/b1.json && python bench.py --cpu-sample 0 --holdout 0 > gpurun_out/b3.json
