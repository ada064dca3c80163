"""Checkpoint interchange with the reference.

The reference writes `{'state_dict', 'optimizer', 'lr', 'steps'}` with `torch.save` (src/main/runner.py:369-371) and
reads checkpoints through `load_checkpoint_with_shape_match` (src/utils/utils.py:352-370): a 'module.' prefix left by
DataParallel is stripped and only entries whose shape matches the model are taken, which is how a model of one variant
is initialised from a checkpoint of another ("transfer learning", args.py:95-100).  camradepth_amd keeps parameters in
the reference's own names, shapes and NCHW fp32 layout (they are views of one flat buffer; the bf16 packed forms the
kernels read are derived every step), so a checkpoint is interchangeable in both directions without conversion."""
import torch


def strip_module_prefix(state_dict):
    return {k.replace("module.", ""): v for k, v in state_dict.items()}


def load_state_dict_shape_match(model, checkpoint_state_dict):
    """utils.py:352-370.  Returns (missing, mismatched): keys of the model absent from the checkpoint and keys whose
    shapes differ (both keep the model's current values); the reference prints them."""
    ckpt = strip_module_prefix(checkpoint_state_dict)
    own = model.state_dict()
    new, missing, mismatched = {}, [], []
    for key, cur in own.items():
        if key in ckpt and tuple(ckpt[key].shape) == tuple(cur.shape):
            new[key] = ckpt[key]
        else:
            (missing if key not in ckpt else mismatched).append(key)
            new[key] = cur
    model.load_state_dict(new, strict=True)
    return missing, mismatched


def save_checkpoint(path, model, optimizer=None, steps=(0, 0)):
    """Same dictionary as runner.py:369 (tensors moved to the CPU; the model itself stays on its device)."""
    state = {"state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "steps": list(steps)}
    if optimizer is not None:
        osd = optimizer.state_dict()
        for st in osd["state"].values():
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.detach().cpu().clone()
        state["optimizer"] = osd
        state["lr"] = optimizer.param_groups[0]["lr"]
    torch.save(state, path)
    return state


def load_checkpoint(path_or_state, model, optimizer=None, shape_match=True):
    """Loads `state_dict` (with the reference's shape-matching rule unless shape_match=False) and, if given and present,
    the optimizer state.  Returns (missing, mismatched, steps)."""
    state = torch.load(path_or_state, map_location="cpu", weights_only=False) if isinstance(path_or_state, str) else path_or_state
    sd = state["state_dict"] if "state_dict" in state else state
    if shape_match:
        missing, mismatched = load_state_dict_shape_match(model, sd)
    else:
        model.load_state_dict(strip_module_prefix(sd))
        missing, mismatched = [], []
    if optimizer is not None and "optimizer" in state:
        optimizer.load_state_dict(state["optimizer"])
    return missing, mismatched, state.get("steps", [0, 0])
