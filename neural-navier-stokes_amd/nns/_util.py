"""Host<->device plumbing shared by the solver mirrors."""
import numpy as np
import torch

TORCH_DTYPE = {np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64}


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("nns needs a HIP device (MI355X); there is no CPU fallback for the solver kernels")
    return torch.device('cuda', torch.cuda.current_device())


def to_dev(x, dtype, device):
    """numpy array or torch tensor -> contiguous device tensor of numpy dtype ``dtype`` (always a copy for
    numpy inputs; device tensors of the right type are passed through)."""
    td = TORCH_DTYPE[np.dtype(dtype)]
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=td).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=dtype), device=device)


def like_input(t, ref):
    """Return device tensor ``t`` in the container type of ``ref`` (numpy in -> numpy float64-or-dtype out)."""
    if isinstance(ref, torch.Tensor):
        return t
    return t.cpu().numpy()
