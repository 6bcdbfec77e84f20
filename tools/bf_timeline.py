"""Per-step timeline stamps (EVMI_F32_TL) of the bf16-operand conv kernel on two dense layers."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
ops.CONV_BACKEND["operands"] = os.environ.get("OPERANDS", "bf16")
def run(cin, cout, k, s, p, d, g, b, t):
    x = torch.randn(cin, b, t, device=dev); w = torch.randn(cout, cin // g, k, device=dev) * 0.1
    os.environ.pop("EVMI_F32_TL", None)
    for _ in range(2): ops.conv1d_mfma(x, w, None, s, p, d, g)
    torch.cuda.synchronize()
    os.environ["EVMI_F32_TL"] = "1"
    ops.conv1d_mfma(x, w, None, s, p, d, g)
    torch.cuda.synchronize()
run(1024, 1024, 5, 1, 2, 1, 1, 16, 128)
run(128, 128, 11, 1, 25, 5, 1, 16, 2048)
