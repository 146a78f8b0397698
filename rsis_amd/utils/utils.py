"""Host-side glue mirroring reference src/utils/utils.py (names, argument meaning and behaviour)."""
import os
import pickle
from collections import OrderedDict

import torch


def make_dir(dir):
    if not os.path.exists(dir):
        os.mkdir(dir)


def get_skip_dims(model_name):
    """reference utils/utils.py:129-137"""
    if model_name == "resnet50" or model_name == "resnet101":
        return [2048, 1024, 512, 256, 64]
    elif model_name == "resnet34":
        return [512, 256, 128, 64, 64]
    elif model_name == "vgg16":
        return [512, 512, 256, 128, 64]
    raise Exception("The base model you chose is not supported !")


def check_parallel(encoder_dict, decoder_dict):
    """reference utils/utils.py:12-32: strip the `module.` prefix of DataParallel-trained checkpoints (the reference
    inspects only the first encoder key)."""
    trained_parallel = False
    for k, _v in encoder_dict.items():
        if k[:7] == "module.":
            trained_parallel = True
        break
    if trained_parallel:
        encoder_dict = OrderedDict((k[7:], v) for k, v in encoder_dict.items())
        decoder_dict = OrderedDict((k[7:], v) for k, v in decoder_dict.items())
    return encoder_dict, decoder_dict


def get_base_params(args, model):
    """reference utils/utils.py:34-52.  The reference walks b[i].modules() and yields j.parameters() for EVERY nested
    module, so tensors are yielded 1-4 times; here each trunk tensor is yielded once --
    use `base_param_multiplicity` to reproduce the reference's effective 1x/3x/4x encoder learning rate."""
    seen = set()
    for m in (model.base.conv1, model.base.bn1, model.base.layer1, model.base.layer2, model.base.layer3, model.base.layer4):
        for p in m.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                yield p


def base_param_multiplicity(model):
    """How many times the reference's get_base_params generator yields each trunk tensor (stem x1, bottleneck
    conv/bn params x3, downsample params x4): {param: count}."""
    counts = {}
    for top in (model.base.conv1, model.base.bn1, model.base.layer1, model.base.layer2, model.base.layer3, model.base.layer4):
        for j in top.modules():
            for k in j.parameters():
                counts[k] = counts.get(k, 0) + 1
    return counts


def get_skip_params(model):
    """reference utils/utils.py:54-71"""
    for m in (model.sk1, model.sk2, model.sk3, model.sk4, model.sk5, model.bn1, model.bn2, model.bn3, model.bn4, model.bn5):
        for p in m.parameters():
            yield p


def merge_params(params):
    for j in range(len(params)):
        for i in params[j]:
            yield i


def get_optimizer(optim_name, lr, parameters, weight_decay=0, momentum=0.9):
    """reference utils/utils.py:78-87 (stock torch optimizers; the fused flat HIP Adam lives in rsis_amd.optim)."""
    params = [p for p in parameters if p.requires_grad]
    if optim_name == "sgd":
        return torch.optim.SGD(params, lr=lr, weight_decay=weight_decay, momentum=momentum)
    elif optim_name == "adam":
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)
    elif optim_name == "rmsprop":
        return torch.optim.RMSprop(params, lr=lr, weight_decay=weight_decay)
    raise Exception("unknown optimizer %s" % optim_name)


def save_checkpoint(args, encoder, decoder, enc_opt, dec_opt, root="../models"):
    """reference utils/utils.py:89-95 (same five files)."""
    d = os.path.join(root, args.model_name)
    os.makedirs(d, exist_ok=True)
    torch.save(encoder.state_dict(), os.path.join(d, "encoder.pt"))
    torch.save(decoder.state_dict(), os.path.join(d, "decoder.pt"))
    torch.save(enc_opt.state_dict(), os.path.join(d, "enc_opt.pt"))
    torch.save(dec_opt.state_dict(), os.path.join(d, "dec_opt.pt"))
    pickle.dump(args, open(os.path.join(d, "args.pkl"), "wb"))


class _ArgsUnpickler(pickle.Unpickler):
    """args.pkl holds an argparse.Namespace of plain values (train.py:234): refuse to resolve anything else, so that loading a
    third-party checkpoint cannot execute code.  "Plain values" includes numpy SCALARS: the reference stores
    `args.best_val_loss = np.mean(...)` (train.py:406-443), a numpy.float64, before every save_checkpoint, whose pickle
    reconstructs through numpy.core.multiarray.scalar(numpy.dtype(...), bytes) -- both allow-listed (they build a scalar from
    raw bytes; object dtypes are refused), and the loaded scalars are turned into python numbers by `_plain`."""
    _ALLOWED = {("argparse", "Namespace"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"),
                ("__builtin__", "dict"), ("__builtin__", "list"), ("__builtin__", "tuple"), ("__builtin__", "set"),
                ("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor"), ("__builtin__", "object"), ("builtins", "object")}
    _NUMPY = {("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "dtype")}
    _ALLOWED = _ALLOWED | {("_codecs", "encode")}      # how a python-3 protocol-2 pickle spells a bytes payload (str -> bytes)

    def find_class(self, module, name):
        if (module, name) in self._NUMPY:
            import numpy as np
            if name == "dtype":
                return _plain_dtype
            return np._core.multiarray.scalar if hasattr(np, "_core") else np.core.multiarray.scalar
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError("args.pkl: refusing to load %s.%s" % (module, name))
        return super().find_class(module, name)


def _plain_dtype(*a, **kw):
    """numpy.dtype for the unpickler: numeric / bool scalars only (an object dtype would make numpy unpickle arbitrary payloads)"""
    import numpy as np
    dt = np.dtype(*a, **kw)
    if dt.kind not in "biuf":
        raise pickle.UnpicklingError("args.pkl: refusing numpy dtype %r" % (dt,))
    return dt


def _plain(v):
    """numpy scalars -> python numbers (recursively through the containers an args namespace holds)"""
    import numpy as np
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, (list, tuple)):
        return type(v)(_plain(x) for x in v)
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    return v


def load_checkpoint(model_name, use_gpu=True, root="../models"):
    """reference utils/utils.py:97-111.  The four .pt files are tensor dictionaries: files in torch's current zip format (this build's
    own checkpoints, anything saved by torch >= 1.6) are read with weights_only=True; files in the pre-zip container the REFERENCE's
    torch 0.2 wrote (python-2 protocol-2 pickles, `torch.cuda.FloatTensor` objects) go through rsis_amd.utils.legacy_pt, a closed-list
    reader that needs neither the GPU they were saved from nor code execution.
    CONTRACT (differs from the reference, whose `torch.load` restores tensors to the device they were saved from): EVERY tensor of the
    four dictionaries, optimizer state included, comes back ON THE CPU whatever `use_gpu` says -- the argument is accepted for
    signature compatibility only.  `load_state_dict` / `FlatGroup.copy_` copy them to wherever the modules live; a caller that uses
    the returned tensors directly must move them itself (`.cuda()`).  args.pkl (python 2 wrote it with
    protocol 0) goes through an allow-listed unpickler, `latin1` for its str payloads."""
    from . import legacy_pt
    d = os.path.join(root, model_name)
    dicts = []
    for f in ("encoder.pt", "decoder.pt", "enc_opt.pt", "dec_opt.pt"):
        path = os.path.join(d, f)
        if legacy_pt.is_legacy_file(path):
            dicts.append(legacy_pt.load(path))
        else:
            dicts.append(torch.load(path, map_location=(lambda storage, location: storage), weights_only=True))
    with open(os.path.join(d, "args.pkl"), "rb") as f:
        try:
            args = _ArgsUnpickler(f).load()
        except UnicodeDecodeError:
            f.seek(0)
            args = _ArgsUnpickler(f, encoding="latin1").load()
    for k, v in list(vars(args).items()):
        setattr(args, k, _plain(v))
    return dicts[0], dicts[1], dicts[2], dicts[3], args


def batch_to_var(args, inputs, targets):
    """reference utils/utils.py:113-127: split the [B, gt_T, H*W+3] target tensor."""
    x = inputs
    y_mask = targets[:, :, :-3].float()
    y_class = targets[:, :, -3].long()
    sw_mask = targets[:, :, -2]
    sw_class = targets[:, :, -1]
    if args.use_gpu:
        return x.cuda(), y_mask.cuda(), y_class.cuda(), sw_mask.cuda(), sw_class.cuda()
    return x, y_mask, y_class, sw_mask, sw_class
