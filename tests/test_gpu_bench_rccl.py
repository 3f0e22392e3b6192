"""-m gpu, needs >= 2 GPUs (skips on the one-GPU test box): bench.py's N > 1 path on RCCL - `torch.distributed.run` with two
ranks on two devices, backend "nccl": the one-time cloud broadcast, the batched vertex-position broadcasts of
multiview.MeshStatePipe on their own stream, view sharding, max-over-ranks timing.  Each rank's last image must equal a
single-process render of ITS view bit for bit (the same check tests/test_gpu_bench_multirank.py makes over gloo on one GPU)."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs two devices")]


@pytest.mark.parametrize("batch", [None, 1])
def test_bench_two_ranks_on_rccl(tmp_path, batch):
    from test_gpu_bench_multirank import _two_ranks
    # no IPC variable here: bench.py itself selects the dmabuf IPC mode RCCL needs on these hosts (HSA_ENABLE_IPC_MODE_LEGACY=0, set
    # before HIP starts), so this runs in the environment the driver's `bench.py --gpus N` runs in
    _two_ranks(tmp_path, batch, dict(GM_BENCH_BACKEND="nccl"))
