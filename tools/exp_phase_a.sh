timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "release_their_device_memory" 2>&1 | tail -5
