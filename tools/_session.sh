export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "bound or reference_tree or msda or lds_plane or 16bit" 2>&1 | tail -8
