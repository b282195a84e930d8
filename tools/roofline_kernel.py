"""Launch ONE of the six matrix-core convolution kernels of bench.py's roofline table a few times, for rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE need separate passes on gfx950) and --kernel-trace --stats.
usage: python3 tools/roofline_kernel.py "<decnn.7|decnn.4> <forward|d/d input|d/d weight>" [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'decnn.7 forward'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
launch = bench.conv_kernel_launchers(B, torch.device('cuda', 0))[name][0]
with torch.no_grad():
    for _ in range(5):
        launch()
torch.cuda.synchronize()
print('done', name, B)
