"""Model discovery and host->device transfer (reference: opencood/tools/train_utils.py:141-174
create_model, :277-286 to_device)."""
import importlib


def create_model(hypes):
    name = hypes['model']['core_method']
    lib = importlib.import_module("heal_amd.opencood.models." + name)
    target = name.replace('_', '').lower()
    for cname, cls in lib.__dict__.items():
        if cname.lower() == target:
            return cls(hypes['model']['args'])
    raise ImportError(f"no class matching '{target}' in heal_amd.opencood.models.{name}")


def create_loss(hypes):
    """train_utils.py:177-210: the class in heal_amd.opencood.loss.<core_method> whose lower-cased name is the
    core_method without underscores, constructed with hypes['loss']['args']."""
    name = hypes['loss']['core_method']
    lib = importlib.import_module("heal_amd.opencood.loss." + name)
    target = name.replace('_', '').lower()
    for cname, cls in lib.__dict__.items():
        if cname.lower() == target:
            return cls(hypes['loss']['args'])
    raise ImportError(f"no class matching '{target}' in heal_amd.opencood.loss.{name}")


def to_device(inputs, device):
    if isinstance(inputs, list):
        return [to_device(x, device) for x in inputs]
    if isinstance(inputs, dict):
        return {k: to_device(v, device) for k, v in inputs.items()}
    if isinstance(inputs, (int, float, str)) or not hasattr(inputs, 'to'):
        return inputs
    return inputs.to(device, non_blocking=True)
