"""Random-shape check of the fused stage-1 kernel (wt_resblock_down) against the oracle: lengths around tile edges and clip
ends, any remainder modulo the stride, small and odd batch sizes.  (A script, not collected by pytest; it lives under tests/
because it uses the oracle, which only tests/, smoke() and bench.py's cpu_baseline leg may import.)  python tests/fuzz_stage1.py [cases per architecture]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(n_cases):
    from wavtokenizer_amd import NAMED_ARCHS, synth
    from wavtokenizer_amd._capi import lib, check
    from oracle.cpu_ref import OracleWavTokenizer
    ENC = "feature_extractor.encodec.encoder.model."
    rng = np.random.default_rng(12345)
    worst = 0.0
    for arch_name, r in (("hop600", 4), ("hop320", 2)):
        arch = NAMED_ARCHS[arch_name]
        sd = synth.make_state_dict(arch, seed=0)
        orc = OracleWavTokenizer(arch, sd)

        def fold(prefix):
            g = torch.from_numpy(sd[prefix + ".weight_g"]).double()
            v = torch.from_numpy(sd[prefix + ".weight_v"]).double()
            w = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
            return w.float().permute(0, 2, 1).contiguous(), torch.from_numpy(sd[prefix + ".bias"])

        w3, b3 = fold(ENC + "1.block.1.conv.conv")
        w1, b1 = fold(ENC + "1.block.3.conv.conv")
        ws, bs = fold(ENC + "1.shortcut.conv.conv")
        wd, bd = fold(ENC + "3.conv.conv")
        e0w, e0b = fold(ENC + "0.conv.conv")
        e0w = e0w[:, :, 0].t().contiguous()
        dev = lambda t: t.cuda().contiguous()
        wts = list(map(dev, (e0w, e0b, w3, b3, w1.reshape(32, 16), b1, ws.reshape(32, 32), bs, wd, bd)))
        opt = (126 - 2 * r) // r + 1
        for case in range(n_cases):
            B = int(rng.integers(1, 6))
            k = int(rng.integers(1, 40))
            T = max(1024, k * opt * r + int(rng.integers(-2 * r - 3, 2 * r + 4)) + int(rng.integers(0, 2)) * 1024)
            wav = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32) * 0.3)
            with torch.inference_mode():
                x = orc.sconv1d(wav.unsqueeze(1), ENC + "0.conv.conv")
                want = orc.sconv1d(F.elu(orc.resblock(x, ENC + "1")), ENC + "3.conv.conv", stride=r).numpy()
            Td = -(-T // r)
            y = torch.full((B, Td, 64), float("nan"), device="cuda")
            wv = dev(wav)
            check(lib.wt_resblock_down(wv.data_ptr(), *[w.data_ptr() for w in wts], y.data_ptr(), B, T, r, None), "wt_resblock_down")
            torch.cuda.synchronize()
            got = y.permute(0, 2, 1).cpu().numpy()
            assert got.shape == want.shape, (got.shape, want.shape)
            err = float(np.sqrt(((got.astype(np.float64) - want) ** 2).sum() / (want.astype(np.float64) ** 2).sum()))
            edge = float(np.abs(got[:, :, -3:] - want[:, :, -3:]).max() / (np.abs(want).max() + 1e-30))
            worst = max(worst, err, edge)
            assert np.isfinite(got).all() and err < 1e-5 and edge < 1e-5, (arch_name, B, T, err, edge)
        print(arch_name, n_cases, "cases ok; worst so far %.2e" % worst, flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
