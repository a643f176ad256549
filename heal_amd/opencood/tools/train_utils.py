"""Model / loss discovery, checkpoint loading, optimizer set-up and host->device transfer (reference:
opencood/tools/train_utils.py:28-102 check_missing_key + load_saved_model, :141-174 create_model, :177-210 create_loss,
:213-274 setup_optimizer + setup_lr_schedular, :277-286 to_device)."""
import glob
import importlib
import os
import re

import torch


def check_missing_key(model_state_dict, ckpt_state_dict):
    """train_utils.py:28-51: report (by top-level module) what a checkpoint lacks or has in excess."""
    missing = set(model_state_dict.keys()) - set(ckpt_state_dict.keys())
    extra = set(ckpt_state_dict.keys()) - set(model_state_dict.keys())
    missing_modules = {k.split('.')[0] for k in missing}
    extra_modules = {k.split('.')[0] for k in extra}
    print("------ Loading Checkpoint ------")
    if not missing_modules and not extra_modules:
        return
    print("Missing keys from ckpt:")
    print(*missing_modules, sep='\n', end='\n\n')
    print("Extra keys from ckpt:")
    print(*extra_modules, sep='\n', end='\n\n')
    print(*extra, sep='\n', end='\n\n')
    print("--------------------------------")


def load_saved_model(saved_path, model):
    """train_utils.py:54-102 (tools/inference.py:98, tools/train.py): prefer `net_epoch_bestval_at<E>.pth` (exactly one
    may exist), else the highest `net_epoch<E>.pth`; `load_state_dict(strict=False)` from a CPU map.  Returns
    (epoch, model); (0, model) when the directory holds no checkpoint."""
    assert os.path.exists(saved_path), '{} not found'.format(saved_path)
    best = glob.glob(os.path.join(saved_path, 'net_epoch_bestval_at*.pth'))
    if best:
        assert len(best) == 1
        epoch = int(re.fullmatch(r"net_epoch_bestval_at(\d+)\.pth", os.path.basename(best[0])).group(1))
        print("resuming best validation model at epoch %d" % epoch)
        state = torch.load(best[0], map_location='cpu')
        check_missing_key(model.state_dict(), state)
        model.load_state_dict(state, strict=False)
        return epoch, model
    epochs = [int(re.findall(".*epoch(.*).pth.*", f)[0]) for f in glob.glob(os.path.join(saved_path, '*epoch*.pth'))]
    epoch = max(epochs) if epochs else 0
    if epoch > 0:
        print('resuming by loading epoch %d' % epoch)
        state = torch.load(os.path.join(saved_path, 'net_epoch%d.pth' % epoch), map_location='cpu')
        check_missing_key(model.state_dict(), state)
        model.load_state_dict(state, strict=False)
    return epoch, model


def setup_optimizer(hypes, model):
    """train_utils.py:213-234: torch.optim.<core_method>(model.parameters(), lr=..., **args)."""
    cfg = hypes['optimizer']
    method = getattr(torch.optim, cfg['core_method'], None)
    if not method:
        raise ValueError('{} is not supported'.format(cfg['name']))
    return method(model.parameters(), lr=cfg['lr'], **cfg.get('args', {}))


def setup_lr_schedular(hypes, optimizer, init_epoch=None):
    """train_utils.py:237-274: 'step' | 'multistep' | exponential, advanced by `init_epoch` steps."""
    from torch.optim import lr_scheduler
    cfg = hypes['lr_scheduler']
    if cfg['core_method'] == 'step':
        scheduler = lr_scheduler.StepLR(optimizer, step_size=cfg['step_size'], gamma=cfg['gamma'])
    elif cfg['core_method'] == 'multistep':
        scheduler = lr_scheduler.MultiStepLR(optimizer, milestones=cfg['step_size'], gamma=cfg['gamma'])
    else:
        scheduler = lr_scheduler.ExponentialLR(optimizer, cfg['gamma'])
    for _ in range(init_epoch if init_epoch is not None else 0):
        scheduler.step()
    return scheduler


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
