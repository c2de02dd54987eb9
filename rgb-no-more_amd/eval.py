"""Mirror of the reference's eval loop reduction (`eval.py:8-51`): top-1 accuracy and mean loss of a DCT model over a
dataloader, summed across the ranks of a process group.

The reference accumulates `torchmetrics.Accuracy(top_k=1)` (correct / total over all ranks) and averages the per-batch
mean loss over the batches of a rank, then over ranks (`dist.reduce(..., AVG)` to rank 0; here every rank gets the
result).  The model call follows `eval.py:37-41`: `model(images[0], images[1])` for the DCT datasets.
"""
import torch
import torch.distributed as dist


@torch.no_grad()
def evaluate_model(model, dataloader, criterion=None, device=None, amp_dtype=None, process_group=None, verbose=False):
    """-> (accuracy, loss).  dataloader yields ((Y, CbCr), labels) or (Y, CbCr, labels); tensors are moved to `device`
    (default: the model's first parameter).  amp_dtype: torch.bfloat16 for autocast (float16, the reference's hard-coded eval dtype, eval.py:36, is executed as bfloat16 by the HIP model), None for fp32."""
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    criterion = criterion if criterion is not None else torch.nn.CrossEntropyLoss()
    was_training = model.training
    model.eval()
    correct = torch.zeros(1, dtype=torch.float64, device=device)
    total = torch.zeros(1, dtype=torch.float64, device=device)
    loss_sum = torch.zeros(1, dtype=torch.float32, device=device)
    nbatch = 0
    for i, data in enumerate(dataloader):
        if len(data) == 2:
            (y, cbcr), labels = data
        else:
            y, cbcr, labels = data
        y, cbcr, labels = y.to(device), cbcr.to(device), labels.to(device)
        if amp_dtype is not None:
            with torch.autocast(device.type, dtype=amp_dtype):
                out = model(y, cbcr)
        else:
            out = model(y, cbcr)
        out = out.float()
        loss_sum += criterion(out, labels).float()
        correct += (out.argmax(dim=1) == labels).sum()
        total += labels.numel()
        nbatch += 1
        if verbose:
            print(f"\rEvaluating... {i + 1}/{len(dataloader)}   ", end="", flush=True)
    loss = loss_sum / max(nbatch, 1)
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(process_group)
        dist.all_reduce(correct, group=process_group)
        dist.all_reduce(total, group=process_group)
        dist.all_reduce(loss, group=process_group)
        loss = loss / world
    if was_training:
        model.train()
    return (correct / total.clamp(min=1)).item(), loss.item()
