import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def randn_bf16(*shape, seed=0, scale=1.0, device="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device=device, dtype=torch.bfloat16)
