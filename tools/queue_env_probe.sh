pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], "fps %.1f" % d["value"])'
python -c "import sys; sys.argv=['bench.py','--no-cpu-baseline','--no-fwd-bwd','--no-c5']; import bench; bench.main()" 2>/dev/null | python -c "$pick" A_env_before_torch
python -c "import sys, torch; sys.argv=['bench.py','--no-cpu-baseline','--no-fwd-bwd','--no-c5']; import bench; bench.main()" 2>/dev/null | python -c "$pick" B_env_after_import_torch
python -c "import sys, torch; torch.cuda.is_available(); sys.argv=['bench.py','--no-cpu-baseline','--no-fwd-bwd','--no-c5']; import bench; bench.main()" 2>/dev/null | python -c "$pick" C_env_after_is_available
python -c "import sys, torch; torch.zeros(1, device='cuda'); sys.argv=['bench.py','--no-cpu-baseline','--no-fwd-bwd','--no-c5']; import bench; bench.main()" 2>/dev/null | python -c "$pick" D_env_after_first_allocation
GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu-baseline --no-fwd-bwd --no-c5 2>/dev/null | python -c "$pick" E_four_queues
