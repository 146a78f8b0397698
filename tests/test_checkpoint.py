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


def _py2_opcodes(raw):
    import pickletools
    return [op.name for op, _arg, _pos in pickletools.genops(raw)]


def test_python2_torch02_byte_streams_load(tmp_path):
    """N4 against the reference environment's actual byte streams (oracle/legacy_ckpt.py emits them opcode by opcode: python 2.7
    cPickle protocol 2 inside torch 0.2's pre-zip container for the four .pt files, python 2 protocol 0 for args.pkl).  Checked:
    the streams hold python-2 opcodes only (py2 `str` = SHORT_BINSTRING, never BINUNICODE; NEWOBJ + BUILD tensors; BINPERSID storages),
    stock torch's own legacy reader accepts the container (CPU-typed variant), and the product path -- load_checkpoint ->
    check_parallel -> load_state_dict -- restores every tensor bit for bit from the `torch.cuda.FloatTensor` variant on a host
    without that device, with the optimizer dictionaries and the args namespace intact."""
    import io
    from oracle import legacy_ckpt as G
    from rsis_amd.utils import legacy_pt
    a = _args(tmp_path, "py2")
    enc, dec = FeatureExtractor(a), RSIS(a)
    _fill(enc, 5)
    _fill(dec, 6)
    ns = {k: v for k, v in vars(a).items() if isinstance(v, (bool, int, float, str, type(None)))}
    ns.update(epoch_resume=7, best_val_loss=np.float64(0.625), use_gpu=True)
    dec_keys = list(dec.state_dict())
    dec_opt = {"state": {94001 + i: {"step": 11, "exp_avg": np.full(tuple(dec.state_dict()[k].shape), 0.5, np.float32),
                                     "exp_avg_sq": np.full(tuple(dec.state_dict()[k].shape), 0.25, np.float32)} for i, k in enumerate(dec_keys)},
               "param_groups": [{"lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "params": [94001 + i for i in range(len(dec_keys))]}]}
    d = G.write_reference_checkpoint(str(tmp_path), "py2", enc.state_dict(), dec.state_dict(), ns, parallel=True, cuda=True, dec_opt=dec_opt)
    # --- the streams are python 2's
    with open(os.path.join(d, "decoder.pt"), "rb") as f:
        raw = f.read()
    assert legacy_pt.is_legacy_file(os.path.join(d, "decoder.pt")) and raw[:2] == b"\x80\x02"
    head = io.BytesIO(raw)
    for _ in range(3):                       # magic number, protocol version, sys_info
        pickle.load(head)
    ops = _py2_opcodes(raw[head.tell():])
    assert "SHORT_BINSTRING" in ops and "NEWOBJ" in ops and "BUILD" in ops and "BINPERSID" in ops and "BINPUT" in ops, sorted(set(ops))
    assert not {"BINUNICODE", "SHORT_BINUNICODE", "BINBYTES", "SHORT_BINBYTES", "FRAME", "MEMOIZE"} & set(ops), sorted(set(ops))
    assert b"ctorch.cuda\nFloatTensor" in raw and b"U\x14module.conv_out.bias" in raw
    with open(os.path.join(d, "args.pkl"), "rb") as f:
        araw = f.read()
    # (pickletools cannot disassemble it: its STRING decoder insists on ASCII and the float64 payload is an escaped byte string)
    assert araw.startswith(b"ccopy_reg\n_reconstructor\n") and b"sS'hidden_size'\n" in araw and b"\x80" not in araw[:2] and b"I01\n" in araw
    assert b"cnumpy.core.multiarray\nscalar\n" in araw
    # --- product path
    e_sd, d_sd, e_opt, d_opt, largs = load_checkpoint("py2", use_gpu=True, root=str(tmp_path))
    assert largs.epoch_resume == 7 and largs.hidden_size == 32 and largs.use_gpu is True and largs.model_name == "py2"
    assert type(largs.best_val_loss) is float and largs.best_val_loss == 0.625
    assert all(k.startswith("module.") and isinstance(k, str) for k in list(e_sd) + list(d_sd))
    assert not any("num_batches_tracked" in k for k in e_sd)
    e_sd, d_sd = check_parallel(e_sd, d_sd)
    enc2, dec2 = FeatureExtractor(largs), RSIS(largs)
    enc2.load_state_dict(e_sd)
    dec2.load_state_dict(d_sd)
    for m, m2 in ((enc, enc2), (dec, dec2)):
        for k, v in m.state_dict().items():
            if "num_batches_tracked" not in k:
                assert torch.equal(v, m2.state_dict()[k]), k
    assert e_opt == {"state": {}, "param_groups": []}
    from rsis_amd.optim import FlatAdam
    opt = FlatAdam(list(dec2.parameters()), lr=1e-3, name="dec")
    assert opt.load_state_dict(d_opt) is not False and opt.group.step_count == 11
    assert float(opt.group.exp_avg.min()) == 0.5 and float(opt.group.exp_avg_sq.max()) == 0.25
    # --- stock torch reads the same container (CPU-typed tensors: this host has no GPU to put torch.cuda.FloatTensor on)
    d2 = G.write_reference_checkpoint(str(tmp_path), "py2cpu", enc.state_dict(), dec.state_dict(), ns, parallel=False, cuda=False)
    stock = torch.load(os.path.join(d2, "decoder.pt"), weights_only=False)
    ours = legacy_pt.load(os.path.join(d2, "decoder.pt"))
    assert list(stock) == list(ours) == dec_keys and all(torch.equal(stock[k], ours[k]) and torch.equal(ours[k], dec.state_dict()[k]) for k in stock)


def test_legacy_reader_refuses_code_and_truncation(tmp_path):
    from oracle import legacy_ckpt as G
    from rsis_amd.utils import legacy_pt
    p = os.path.join(str(tmp_path), "t.pt")
    G.torch02_save(OrderedDict([("w", np.arange(12, dtype=np.float32).reshape(3, 4))]), p)
    assert torch.equal(legacy_pt.load(p)["w"], torch.arange(12.0).reshape(3, 4))
    with open(p, "rb") as f:
        raw = f.read()
    with open(p, "wb") as f:
        f.write(raw[:-8])                                    # storage cut short
    try:
        legacy_pt.load(p)
        raise AssertionError("truncated storage accepted")
    except pickle.UnpicklingError:
        pass
    evil = raw.replace(b"ccollections\nOrderedDict\n", b"cos\nsystem\n", 1)
    with open(p, "wb") as f:
        f.write(evil)
    try:
        legacy_pt.load(p)
        raise AssertionError("foreign global resolved")
    except pickle.UnpicklingError as e:
        assert "os.system" in str(e)
    big = raw.replace(b"K\x03K\x04\x86", b"K\x09K\x04\x86", 1)   # size (3, 4) -> (9, 4): reaches outside its 12-element storage
    assert big != raw
    with open(p, "wb") as f:
        f.write(big)
    try:
        legacy_pt.load(p)
        raise AssertionError("out-of-storage tensor accepted")
    except pickle.UnpicklingError:
        pass


def test_legacy_reader_reads_bf16_and_bool_storages(tmp_path):
    """a pre-zip container written by a CURRENT torch (`_use_new_zipfile_serialization=False`) may hold BFloat16Storage /
    BoolStorage, which torch 0.2 did not have: bf16 is read as 16-bit words and re-viewed, bool as bytes; views and strides as before."""
    from rsis_amd.utils import legacy_pt
    d = {"w": torch.randn(3, 4).bfloat16(), "m": torch.tensor([True, False, True]), "v": torch.randn(8).bfloat16()[2:6],
         "t": torch.arange(6.0).reshape(2, 3).t()}
    p = str(tmp_path / "x.pt")
    torch.save(d, p, _use_new_zipfile_serialization=False)
    assert legacy_pt.is_legacy_file(p)
    got = legacy_pt.load(p)
    for k, v in d.items():
        assert got[k].dtype == v.dtype and got[k].device.type == "cpu" and torch.equal(got[k], v), k
