"""Checkpoint interop with the reference (reference src/utils/utils.py:12-32,89-111): a
reference-era checkpoint directory -- five files, `module.`-prefixed keys when trained under nn.DataParallel, torch-0.2 BN
dicts without `num_batches_tracked`, args pickled by python 2 (protocol 2, str -> bytes under python 3) -- must load into
the build's modules unchanged, and the build's own checkpoints must round-trip.  CPU only: no kernel is launched."""
import argparse
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch

from rsis_amd.args import get_parser
from rsis_amd.modules import RSIS, FeatureExtractor
from rsis_amd.utils.utils import check_parallel, load_checkpoint, save_checkpoint


def _args(tmp, name="ckpt"):
    a = get_parser().parse_args(["-model_name", name, "-hidden_size", "32", "-num_classes", "7"])
    a.models_root = str(tmp)
    a.use_gpu = False
    return a


def _fill(module, seed):
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for _k, v in module.state_dict().items():
            if v.dtype.is_floating_point:
                v.copy_(torch.from_numpy(rng.normal(0, 1, tuple(v.shape)).astype(np.float32)))


def test_reference_era_checkpoint_loads(tmp_path):
    a = _args(tmp_path, "ref_era")
    enc, dec = FeatureExtractor(a), RSIS(a)
    _fill(enc, 1)
    _fill(dec, 2)
    # what the reference wrote: DataParallel prefix, no num_batches_tracked, optimizer dicts, python-2 pickle of the args
    enc_sd = OrderedDict(("module." + k, v.clone()) for k, v in enc.state_dict().items() if "num_batches_tracked" not in k)
    dec_sd = OrderedDict(("module." + k, v.clone()) for k, v in dec.state_dict().items())
    assert any("num_batches_tracked" in k for k in enc.state_dict())       # (the build itself has the modern keys)
    d = os.path.join(str(tmp_path), "ref_era")
    os.makedirs(d)
    torch.save(enc_sd, os.path.join(d, "encoder.pt"))
    torch.save(dec_sd, os.path.join(d, "decoder.pt"))
    torch.save({"state": {}, "param_groups": []}, os.path.join(d, "enc_opt.pt"))
    torch.save({"state": {}, "param_groups": []}, os.path.join(d, "dec_opt.pt"))
    ns = argparse.Namespace(**{k: v for k, v in vars(a).items()})
    ns.epoch_resume = 3
    ns.best_val_loss = np.mean([0.5, 0.75])     # what the reference stores before every save (train.py:406-443): a numpy.float64
    assert isinstance(ns.best_val_loss, np.float64)
    with open(os.path.join(d, "args.pkl"), "wb") as f:
        pickle.dump(ns, f, protocol=2)
    e_sd, d_sd, e_opt, d_opt, largs = load_checkpoint("ref_era", use_gpu=False, root=str(tmp_path))
    assert largs.epoch_resume == 3 and largs.hidden_size == 32
    assert type(largs.best_val_loss) is float and largs.best_val_loss == 0.625
    e_sd, d_sd = check_parallel(e_sd, d_sd)
    assert not any(k.startswith("module.") for k in list(e_sd) + list(d_sd))
    enc2, dec2 = FeatureExtractor(largs), RSIS(largs)
    enc2.load_state_dict(e_sd)          # torch-0.2 BN dicts (no num_batches_tracked) are accepted
    dec2.load_state_dict(d_sd)
    for k, v in enc.state_dict().items():
        if "num_batches_tracked" not in k:
            assert torch.equal(v, enc2.state_dict()[k]), k
    for k, v in dec.state_dict().items():
        assert torch.equal(v, dec2.state_dict()[k]), k
    # the reference weight layout is what is serialised: Gates = [4*hid, in+hid, k, k], gate order i,f,o,g (clstm.py:17,47)
    assert tuple(dec2.state_dict()["clstm_list.0.Gates.weight"].shape) == (4 * 32, 32 + 32, 3, 3)
    assert set(dec2.state_dict()) == {"clstm_list.%d.Gates.%s" % (i, p) for i in range(5) for p in ("weight", "bias")} | {
        "conv_out.weight", "conv_out.bias", "fc_class.weight", "fc_class.bias", "fc_stop.weight", "fc_stop.bias"}


def test_own_checkpoint_roundtrip(tmp_path):
    from rsis_amd.optim import FlatAdam
    a = _args(tmp_path, "own")
    enc, dec = FeatureExtractor(a), RSIS(a)
    _fill(enc, 3)
    _fill(dec, 4)
    enc_opt = FlatAdam(list(enc.base.parameters()), lr=1e-6, name="enc")
    dec_opt = FlatAdam(list(dec.parameters()), lr=1e-3, name="dec")
    dec_opt.group.exp_avg.normal_()
    dec_opt.group.exp_avg_sq.uniform_()
    dec_opt.group.step_count = 5
    save_checkpoint(a, enc, dec, enc_opt, dec_opt, root=str(tmp_path))
    e_sd, d_sd, e_o, d_o, largs = load_checkpoint("own", use_gpu=False, root=str(tmp_path))
    enc2, dec2 = FeatureExtractor(largs), RSIS(largs)
    enc2.load_state_dict(e_sd)
    dec2.load_state_dict(d_sd)
    for k, v in dec.state_dict().items():
        assert torch.equal(v, dec2.state_dict()[k]), k
    dec_opt2 = FlatAdam(list(dec2.parameters()), lr=1e-3, name="dec")
    dec_opt2.load_state_dict(d_o)
    assert torch.equal(dec_opt2.group.exp_avg, dec_opt.group.exp_avg)
    assert torch.equal(dec_opt2.group.exp_avg_sq, dec_opt.group.exp_avg_sq)
    assert dec_opt2.group.step_count == 5


def test_torch_adam_state_is_adopted():
    """the reference's dec_opt.pt is a torch.optim.Adam state_dict (utils/utils.py:93-94, train.py:236-240)"""
    from rsis_amd.optim import FlatAdam
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    ref_opt = torch.optim.Adam(lin.parameters(), lr=1e-3)
    for _ in range(3):
        ref_opt.zero_grad()
        lin(torch.randn(6, 5)).square().mean().backward()
        ref_opt.step()
    sd = ref_opt.state_dict()
    lin2 = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    opt = FlatAdam(lin2.parameters(), lr=1e-3)
    opt.load_state_dict(sd)
    assert opt.group.step_count == 3
    ref_m = torch.cat([sd["state"][i]["exp_avg"].reshape(-1) for i in sd["param_groups"][0]["params"]])
    assert torch.equal(opt.group.exp_avg, ref_m)
    # a parameter list that does not line up (the reference's repeated trunk tensors): moments restart, no exception
    opt3 = FlatAdam(list(lin2.parameters())[:2], lr=1e-3)
    opt3.load_state_dict(sd)
    assert float(opt3.group.exp_avg.abs().sum()) == 0.0


def test_args_pickle_numpy_scalars_and_refusals(tmp_path):
    """args.pkl as python 2 + old numpy wrote it (module path numpy.core.multiarray, str payload -> latin1 under python 3) loads with
    its numpy.float64 turned into a float; anything that is not a plain value or a numeric numpy scalar is refused"""
    from rsis_amd.utils.utils import _ArgsUnpickler
    import io
    # protocol-2 pickle of Namespace(best_val_loss=np.float64(0.625), epoch_resume=np.int64(4)) with the python-2 era module path
    ns = argparse.Namespace(best_val_loss=np.float64(0.625), epoch_resume=np.int64(4), lr=1e-3)
    raw = pickle.dumps(ns, protocol=2).replace(b"numpy._core.multiarray", b"numpy.core.multiarray")
    raw = raw.replace(b"cnumpy._core.multiarray", b"cnumpy.core.multiarray")
    got = _ArgsUnpickler(io.BytesIO(raw)).load()
    assert float(got.best_val_loss) == 0.625 and int(got.epoch_resume) == 4

    class Evil(object):
        def __reduce__(self):
            return (os.system, ("true",))
    for bad in (argparse.Namespace(x=Evil()), argparse.Namespace(x=np.zeros(3)), argparse.Namespace(x=np.array([None], dtype=object)[0:1])):
        try:
            _ArgsUnpickler(io.BytesIO(pickle.dumps(bad, protocol=2))).load()
        except pickle.UnpicklingError:
            continue
        raise AssertionError("the args unpickler accepted %r" % (bad,))
