import os, sys
if len(sys.argv) > 1 and sys.argv[1] == "early":
    os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = "1"
import torch
if len(sys.argv) > 1 and sys.argv[1] == "late":
    os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = "1"
import torch.nn.functional as F
x = torch.randn(4, 64, 128, 128, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 64, 3, 3, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
y = F.conv2d(x, w, None, 1, 1)
print(sys.argv[1:], "out channels_last:", y.is_contiguous(memory_format=torch.channels_last), "contig:", y.is_contiguous(), torch.__version__)
