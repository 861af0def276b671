"""The `spconv` alias the model code imports (reference pcdet/utils/spconv_utils.py:3-34), bound
to the MI355X-native operator namespace, plus the two helpers the reference defines next to it."""
import torch.nn as nn

from ... import spconv  # noqa: F401  (toda_amd.spconv)


def find_all_spconv_keys(model: nn.Module, prefix=""):
    """state_dict keys of every sparse-conv weight (their layout differs between spconv 1.x / 2.x)."""
    keys = set()
    for name, child in model.named_children():
        path = f"{prefix}.{name}" if prefix else name
        if isinstance(child, spconv.conv.SparseConvolution):
            keys.add(f"{path}.weight")
        keys |= find_all_spconv_keys(child, prefix=path)
    return keys


def replace_feature(out, new_features):
    if hasattr(out, "replace_feature"):
        return out.replace_feature(new_features)
    out.features = new_features
    return out
