"""Global device selector with the reference's interface (/root/reference/impl/config.py:6-19):
`set_device(idx)`; idx == -1 or 'cpu' selects the CPU, anything else `cuda:idx` (on ROCm that is
the HIP device) when a GPU is visible.  The models in this package run on the GPU only; a CPU
device makes them raise GlassHipError at the first forward (no silent fallback)."""
import torch

device = None
device_index = None


def set_device(idx):
    global device_index, device
    if idx == 'cpu' or idx == -1:
        device = torch.device('cpu')
    else:
        device_index = idx
        device = torch.device(f'cuda:{device_index}' if torch.cuda.is_available() else 'cpu')
    print("device=", device)
