import sys, re
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
src = open("/root/repo/tests/test_gpu_blk_dec.py").read()
# turn the asserts of the sequence test into prints
i = src.index("def test_blk_decoder_sequence_against")
body = src[i:]
body = body.replace('        assert e_store <= 0.06, "%s: blk storage is %.3f relative L2 from the fp32-storage path" % (name, e_store)\n', '        print("%-34s store %.4f  fp32-storage vs f64 %.4f  blk vs f64 %.4f" % (name, e_store, e0, e1))\n')
body = body.replace('        assert e1 <= 1.5 * e0 + 0.01, "%s: %.4f from float64 (fp32 storage: %.4f)" % (name, e1, e0)\n', '')
body = re.sub(r'        assert float\(\(got - ref\).*\n', '        print("step", t, "out max err / max", float((got - ref).abs().max() / ref.abs().max()))\n', body)
ns = {}
exec(src[:i].replace("pytestmark = pytest.mark.gpu", "") + body, ns)
for g in ("224", "odd"):
    print("=== geometry", g)
    ns["test_blk_decoder_sequence_against_the_fp32_storage_decoder_and_float64"].__wrapped__(g) if hasattr(ns["test_blk_decoder_sequence_against_the_fp32_storage_decoder_and_float64"], "__wrapped__") else ns["test_blk_decoder_sequence_against_the_fp32_storage_decoder_and_float64"](g)
