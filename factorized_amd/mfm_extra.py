"""The reference's ablation and missing-modality model classes on the MI355X HIP ops (drop-in mirror of the rest of
`mfm_model.py`'s class surface):

    M_A, M_B, M_C, M_D             reference mfm_model.py:201-467   ablations of the factorization
    MFM_missing                    reference mfm_model.py:766-898   surrogate encoders for a missing modality
    seq2seq, basic_missing         reference mfm_model.py:900-1017  baselines of the missing-modality study

Same constructors (six dicts), sub-module / parameter names (`state_dict()` keys equal the reference's, checked in
tests/test_module_surface.py) and forward contracts.  They are re-wirings of the blocks `mfm_model.py` already runs on
libmfm_hip.so -- `encoderLSTM`, `decoderLSTM`, `MFN`, `HipLinear`, `loss_MMD` -- so they are COMPOSED from those
autograd ops here, with independent LSTMs / Linears sharing launches (`seq_group`, `decoder_group`, `linear_group`).
There is no fused one-call plan for these classes ("pure re-wiring of existing ops").
"""
import torch
import torch.nn as nn

from . import mfm_model as M
from .mfm_model import HipLinear, MFN, decoderLSTM, encoderLSTM, loss_MMD


class _Base(nn.Module):
    def __init__(self, config):
        super().__init__()
        [self.d_l, self.d_a, self.d_v] = config["input_dims"]
        self.mmd_gauss = None       # optional injected N(0,1) samples, in loss_MMD call order (parity tests)
        self._gi = 0

    def _split(self, x):
        return x[:, :, :self.d_l], x[:, :, self.d_l:self.d_l + self.d_a], x[:, :, self.d_l + self.d_a:]

    def _mmd(self, z):
        g = self.mmd_gauss[self._gi] if self.mmd_gauss is not None else None
        self._gi += 1
        return loss_MMD(z, g)

    def _zf(self, tag, cfg, zin, fout):
        setattr(self, tag + "_fc1", HipLinear(zin, fout))
        setattr(self, tag + "_fc2", HipLinear(fout, fout))
        setattr(self, tag + "_dropout", nn.Dropout(cfg[tag + "_dropout"]))

    def _z_to_f_many(self, pairs):
        """[(tag, z)] -> [relu(fc2(drop(relu(fc1(z)))))], the fc1s in one launch and the fc2s in one launch"""
        h1 = M.linear_group([(z, getattr(self, tag + "_fc1")) for tag, z in pairs])
        h1 = [getattr(self, tag + "_dropout")(torch.relu(v)) for (tag, _), v in zip(pairs, h1)]
        out = M.linear_group([(v, getattr(self, tag + "_fc2")) for (tag, _), v in zip(pairs, h1)])
        return [torch.relu(v) for v in out]

    def _classify(self, f):
        return self.fy_to_y_fc2(self.fy_to_y_dropout(torch.relu(self.fy_to_y_fc1(f))))

    def _classifier(self, cfg, fin, fy):
        self.fy_to_y_fc1 = HipLinear(fin, fy)
        self.fy_to_y_fc2 = HipLinear(fy, cfg["output_dim"])
        self.fy_to_y_dropout = nn.Dropout(cfg["fy_to_y_dropout"])


def _sizes(c):
    return (c["zy_size"], c["zl_size"], c["za_size"], c["zv_size"], c["fy_size"], c["fl_size"], c["fa_size"], c["fv_size"])


def _encode(pairs):
    """[(x, encoderLSTM)] -> [fc1(h_T)] in shared launches"""
    out, _ = M.seq_group(pairs, [])
    return out


class M_A(_Base):
    """reference mfm_model.py:201-266: one early-fusion encoder for z_l, MFN for z_y, all decoders fed [f_y, f_l]."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(M_A, self).__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        last = sum(config["h_dims"]) + config["memsize"]
        self.encoder_l = encoderLSTM(self.d_l + self.d_a + self.d_v, zl)
        self.decoder_l = decoderLSTM(fy + fl, self.d_l)
        self.decoder_a = decoderLSTM(fy + fl, self.d_a)
        self.decoder_v = decoderLSTM(fy + fl, self.d_v)
        self.mfn_encoder = MFN(config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig)
        self.last_to_zy_fc1 = HipLinear(last, zy)
        self._zf("zy_to_fy", config, zy, fy)
        self._zf("zl_to_fl", config, zl, fl)
        self._classifier(config, fy, fy)

    def forward(self, x):
        M._require_cuda(x, "M_A.forward")
        self._gi = 0
        t = x.shape[0]
        zl = self.encoder_l.forward(x)
        zy = self.last_to_zy_fc1(self.mfn_encoder.forward(x))
        mmd_loss = self._mmd(zl) + self._mmd(zy)
        fy, fl = self._z_to_f_many([("zy_to_fy", zy), ("zl_to_fl", zl)])
        fyfl = torch.cat([fy, fl], dim=1)
        x_l_hat, x_a_hat, x_v_hat = M.decoder_group([(fyfl, self.decoder_l), (fyfl, self.decoder_a), (fyfl, self.decoder_v)], t)
        return [x_l_hat, x_a_hat, x_v_hat, self._classify(fy)], mmd_loss, 0.0


class M_B(_Base):
    """reference mfm_model.py:268-335: modality-specific factors only; the classifier reads [f_l, f_a, f_v]."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(M_B, self).__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        self.encoder_l = encoderLSTM(self.d_l, zl)
        self.encoder_a = encoderLSTM(self.d_a, za)
        self.encoder_v = encoderLSTM(self.d_v, zv)
        self.decoder_l = decoderLSTM(fl, self.d_l)
        self.decoder_a = decoderLSTM(fa, self.d_a)
        self.decoder_v = decoderLSTM(fv, self.d_v)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)
        self._classifier(config, fl + fa + fv, fy)

    def forward(self, x):
        M._require_cuda(x, "M_B.forward")
        self._gi = 0
        t = x.shape[0]
        x_l, x_a, x_v = self._split(x)
        zl, za, zv = _encode([(x_l, self.encoder_l), (x_a, self.encoder_a), (x_v, self.encoder_v)])
        mmd_loss = self._mmd(zl) + self._mmd(za) + self._mmd(zv)
        fl, fa, fv = self._z_to_f_many([("zl_to_fl", zl), ("za_to_fa", za), ("zv_to_fv", zv)])
        x_l_hat, x_a_hat, x_v_hat = M.decoder_group([(fl, self.decoder_l), (fa, self.decoder_a), (fv, self.decoder_v)], t)
        y_hat = self._classify(torch.cat([fl, fa, fv], dim=1))
        return [x_l_hat, x_a_hat, x_v_hat, y_hat], mmd_loss, 0.0


class M_C(_Base):
    """reference mfm_model.py:337-387: the multimodal factor only (MFN -> z_y -> f_y feeds every decoder)."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(M_C, self).__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        last = sum(config["h_dims"]) + config["memsize"]
        self.decoder_l = decoderLSTM(fy, self.d_l)
        self.decoder_a = decoderLSTM(fy, self.d_a)
        self.decoder_v = decoderLSTM(fy, self.d_v)
        self.mfn_encoder = MFN(config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig)
        self.last_to_zy_fc1 = HipLinear(last, zy)
        self._zf("zy_to_fy", config, zy, fy)
        self._classifier(config, fy, fy)

    def forward(self, x):
        M._require_cuda(x, "M_C.forward")
        self._gi = 0
        t = x.shape[0]
        zy = self.last_to_zy_fc1(self.mfn_encoder.forward(x))
        mmd_loss = self._mmd(zy)
        (fy,) = self._z_to_f_many([("zy_to_fy", zy)])
        x_l_hat, x_a_hat, x_v_hat = M.decoder_group([(fy, self.decoder_l), (fy, self.decoder_a), (fy, self.decoder_v)], t)
        return [x_l_hat, x_a_hat, x_v_hat, self._classify(fy)], mmd_loss, 0.0


class M_D(_Base):
    """reference mfm_model.py:389-467: purely discriminative (no decoders; `decoded` carries the inputs back)."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(M_D, self).__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        self.encoder_l = encoderLSTM(self.d_l, zl)
        self.encoder_a = encoderLSTM(self.d_a, za)
        self.encoder_v = encoderLSTM(self.d_v, zv)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)
        self.fs_to_y = HipLinear(fl + fa + fv, config["output_dim"])

    def forward(self, x):
        M._require_cuda(x, "M_D.forward")
        x_l, x_a, x_v = self._split(x)
        zl, za, zv = _encode([(x_l, self.encoder_l), (x_a, self.encoder_a), (x_v, self.encoder_v)])
        fl, fa, fv = self._z_to_f_many([("zl_to_fl", zl), ("za_to_fa", za), ("zv_to_fv", zv)])
        y_hat = self.fs_to_y(torch.cat([fl, fa, fv], dim=1))
        return [x_l, x_a, x_v, y_hat], 0.0, 0.0


class MFM_missing(_Base):
    """reference mfm_model.py:766-898: MFM plus six surrogate encoders that predict a modality's (and z_y's) code
    from the other two modalities; forward returns the four decodings (all present / no l / no a / no v)."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(MFM_missing, self).__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        d_l, d_a, d_v = self.d_l, self.d_a, self.d_v
        last = sum(config["h_dims"]) + config["memsize"]
        self.encoder_l = encoderLSTM(d_l, zl)
        self.encoder_a = encoderLSTM(d_a, za)
        self.encoder_v = encoderLSTM(d_v, zv)
        self.encoder_la_to_v = encoderLSTM(d_l + d_a, zv)
        self.encoder_lv_to_a = encoderLSTM(d_l + d_v, za)
        self.encoder_av_to_l = encoderLSTM(d_a + d_v, zl)
        self.encoder_la_to_y = encoderLSTM(d_l + d_a, zy)
        self.encoder_lv_to_y = encoderLSTM(d_l + d_v, zy)
        self.encoder_av_to_y = encoderLSTM(d_a + d_v, zy)
        self.decoder_l = decoderLSTM(fy + fl, d_l)
        self.decoder_a = decoderLSTM(fy + fa, d_a)
        self.decoder_v = decoderLSTM(fy + fv, d_v)
        self.mfn_encoder = MFN(config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig)
        self.last_to_zy_fc1 = HipLinear(last, zy)
        self._zf("zy_to_fy", config, zy, fy)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)
        self._classifier(config, fy, fy)

    def forward(self, x):
        M._require_cuda(x, "MFM_missing.forward")
        self._gi = 0
        t = x.shape[0]
        x_l, x_a, x_v = self._split(x)
        la = x[:, :, :self.d_l + self.d_a]                       # cat([x_l, x_a]) is a contiguous column range
        av = x[:, :, self.d_l:]
        lv = torch.cat([x_l, x_v], dim=2)
        mfn = self.mfn_encoder
        # nine sequence encoders + the MFN's three LSTMs: shared projection / recurrence / head launches
        encs, states = M.seq_group(
            [(x_l, self.encoder_l), (x_a, self.encoder_a), (x_v, self.encoder_v),
             (la, self.encoder_la_to_v), (lv, self.encoder_lv_to_a), (av, self.encoder_av_to_l),
             (la, self.encoder_la_to_y), (lv, self.encoder_lv_to_y), (av, self.encoder_av_to_y)],
            [(x_l, mfn.lstm_l), (x_a, mfn.lstm_a), (x_v, mfn.lstm_v)])
        zl, za, zv, zv_nov, za_noa, zl_nol, zy_nov, zy_noa, zy_nol = encs
        zy = self.last_to_zy_fc1(mfn.forward(x, states))
        mmd_loss = self._mmd(zl) + self._mmd(za) + self._mmd(zv) + self._mmd(zy)
        mse = torch.nn.functional.mse_loss
        missing_loss = mse(zv_nov, zv) + mse(za_noa, za) + mse(zl_nol, zl) + mse(zy_nov, zy) + mse(zy_noa, zy) + mse(zy_nol, zy)

        def decode(zl, za, zv, zy):
            fy, fl, fa, fv = self._z_to_f_many([("zy_to_fy", zy), ("zl_to_fl", zl), ("za_to_fa", za), ("zv_to_fv", zv)])
            x_l_hat, x_a_hat, x_v_hat = M.decoder_group([(torch.cat([fy, fl], dim=1), self.decoder_l),
                                                         (torch.cat([fy, fa], dim=1), self.decoder_a),
                                                         (torch.cat([fy, fv], dim=1), self.decoder_v)], t)
            return [x_l_hat, x_a_hat, x_v_hat, self._classify(fy)]
        decoded = decode(zl, za, zv, zy)
        decoded_nol = decode(zl_nol, za, zv, zy_nol)
        decoded_noa = decode(zl, za_noa, zv, zy_noa)
        decoded_nov = decode(zl, za, zv_nov, zy_nov)
        return decoded, decoded_nol, decoded_noa, decoded_nov, mmd_loss, missing_loss


class seq2seq(_Base):
    """reference mfm_model.py:900-960: reconstruct each modality from the other two (no multimodal factor)."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(seq2seq, self).__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        d_l, d_a, d_v = self.d_l, self.d_a, self.d_v
        self.encoder_la_to_v = encoderLSTM(d_l + d_a, zv)
        self.encoder_lv_to_a = encoderLSTM(d_l + d_v, za)
        self.encoder_av_to_l = encoderLSTM(d_a + d_v, zl)
        self.decoder_l = decoderLSTM(fl, d_l)
        self.decoder_a = decoderLSTM(fa, d_a)
        self.decoder_v = decoderLSTM(fv, d_v)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)

    def forward(self, x):
        M._require_cuda(x, "seq2seq.forward")
        self._gi = 0
        t = x.shape[0]
        x_l, x_a, x_v = self._split(x)
        la, av, lv = x[:, :, :self.d_l + self.d_a], x[:, :, self.d_l:], torch.cat([x_l, x_v], dim=2)
        zv_nov, za_noa, zl_nol = _encode([(la, self.encoder_la_to_v), (lv, self.encoder_lv_to_a), (av, self.encoder_av_to_l)])
        mmd_loss = self._mmd(zv_nov) + self._mmd(za_noa) + self._mmd(zl_nol)
        fl, fa, fv = self._z_to_f_many([("zl_to_fl", zl_nol), ("za_to_fa", za_noa), ("zv_to_fv", zv_nov)])
        x_l_hat, x_a_hat, x_v_hat = M.decoder_group([(fl, self.decoder_l), (fa, self.decoder_a), (fv, self.decoder_v)], t)
        return [x_l_hat], [x_a_hat], [x_v_hat], mmd_loss


class basic_missing(_Base):
    """reference mfm_model.py:962-1017: predict the label from two modalities, one small head per missing modality."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(basic_missing, self).__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        d_l, d_a, d_v = self.d_l, self.d_a, self.d_v
        self.encoder_la_to_y = encoderLSTM(d_l + d_a, zy)
        self.encoder_lv_to_y = encoderLSTM(d_l + d_v, zy)
        self.encoder_av_to_y = encoderLSTM(d_a + d_v, zy)
        for tag in ("zy_nol_to_y", "zy_noa_to_y", "zy_nov_to_y"):
            setattr(self, tag + "_fc1", HipLinear(zy, fy))
            setattr(self, tag + "_fc2", HipLinear(fy, config["output_dim"]))
            setattr(self, tag + "_dropout", nn.Dropout(config["zy_to_fy_dropout"]))

    def forward(self, x):
        M._require_cuda(x, "basic_missing.forward")
        self._gi = 0
        x_l, x_a, x_v = self._split(x)
        la, av, lv = x[:, :, :self.d_l + self.d_a], x[:, :, self.d_l:], torch.cat([x_l, x_v], dim=2)
        zy_nov, zy_noa, zy_nol = _encode([(la, self.encoder_la_to_y), (lv, self.encoder_lv_to_y), (av, self.encoder_av_to_y)])
        mmd_loss = self._mmd(zy_nov) + self._mmd(zy_noa) + self._mmd(zy_nol)
        tags, zs = ("zy_nol_to_y", "zy_noa_to_y", "zy_nov_to_y"), (zy_nol, zy_noa, zy_nov)
        h = M.linear_group([(z, getattr(self, t_ + "_fc1")) for t_, z in zip(tags, zs)])
        h = [getattr(self, t_ + "_dropout")(torch.relu(v)) for t_, v in zip(tags, h)]
        y_hat_nol, y_hat_noa, y_hat_nov = M.linear_group([(v, getattr(self, t_ + "_fc2")) for t_, v in zip(tags, h)])
        return y_hat_nol, y_hat_noa, y_hat_nov, mmd_loss
