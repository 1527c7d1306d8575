#!/usr/bin/env python3
"""Python-3 restatement of the reference's driver loops on the MI355X path.

The reference README tells users to run `mfm_test_mosi.py --config configs/mosi.json` (README.md:36); the nearest real
files are the Python-2 `mfm_mosi.py` (regression datasets) and `mfm_you.py` (classification datasets).  This script
keeps their shape -- argparse `--config`, `seqlength` read from the JSON (mfm_mosi.py:33-47), one shuffle with numpy
seed 123 (:387-389), floor-division batch count (:423), Adam defaults (:403), ReduceLROnPlateau('min') on the
validation loss (:417,472), best-checkpoint rule with a whole-module `torch.save(model, path)` / `torch.load`
(:473-481), the `epoch train valid` log line (:476-479), score() (:483-499 / mfm_you.py:556-564) -- but trains on
synthetic data of the dataset's shape (the CMU pickles are private) with the fused MI355X step:

    --model kl_ef | kl | mmd     MFM_KL_EF / MFM_KL / MFM (train_mfm picks by config['type'], mfm_mosi.py:398-401)
    --staged                     train_beta_vae's schedule (mfm_mosi.py:225-361): `epochs` of gen+reg, then of disc+reg
    --task ce                    mfm_you.py: 3-way cross-entropy head on the YouTube/POM shape
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, metrics, synth  # noqa: E402
from factorized_amd.mfm_model import MFM, MFM_KL, MFM_KL_EF  # noqa: E402
from factorized_amd._lib import MfmError  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--n-train", type=int, default=1280)
    ap.add_argument("--model", choices=["kl_ef", "kl", "mmd"], default="kl_ef")
    ap.add_argument("--task", choices=["l1", "ce"], default="l1",
                    help="l1: regression drivers (mfm_mosi.py); ce: classification drivers (mfm_you.py:451-489)")
    ap.add_argument("--staged", action="store_true")
    ap.add_argument("--legacy-adam", action="store_true",
                    help="staged training with PyTorch-0.4 optimizer semantics (DESIGN.md section 2: parameters without a "
                         "gradient keep moving on their decaying momentum) instead of torch >= 2 (frozen)")
    ap.add_argument("--ckpt", default=None, help="where the best model is saved (whole module); default: a temp file")
    args = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    if args.config is None:
        args.config = os.path.join(here, "..", "configs", "you.json" if args.task == "ce" else "mosi.json")
    _, T = C.load_json_config(args.config)                  # only `seqlength` is read, as in the reference
    cfgs = (C.you_configs if args.task == "ce" else C.canonical_configs)(dropout=True)
    cfg = cfgs[0]
    ce = args.task == "ce"
    classes = cfg["output_dim"] if ce else 0
    np.random.seed(123)
    mk = lambda n, seed: synth.make_dataset(cfg["input_dims"], n, T, seed=seed, output_dim=cfg["output_dim"], classes=classes)
    (Xtr, ytr), (Xva, yva), (Xte, yte) = mk(args.n_train, 11), mk(229, 12), mk(686, 13)
    p = np.random.permutation(Xtr.shape[1])
    Xtr, ytr = Xtr[:, p], ytr[p]
    dev = torch.device("cuda")
    model = {"kl_ef": MFM_KL_EF, "kl": MFM_KL, "mmd": MFM}[args.model](*cfgs).to(dev)
    eng = model.engine                                     # fused step on the module's own storage
    if args.legacy_adam:
        eng.staged_adam = "legacy"
    lr = 1e-3
    best = 999999.0                                        # best-valid rule of the checkpoint (:473; per stage :348)
    # ReduceLROnPlateau(optimizer, 'min') defaults (:253, :417; stepped at :341 / :351 / :472): its OWN best value and
    # bad-epoch counter, kept across both stages; "better" = below best * (1 - 1e-4)
    sched_best, bad, factor, patience = float("inf"), 0, 0.1, 10
    bs = cfg["batchsize"]
    nb = Xtr.shape[1] // bs                                # floor division: the tail is dropped (:423)
    Xd = torch.from_numpy(np.ascontiguousarray(Xtr[:, :nb * bs].reshape(T, nb, bs, -1).transpose(1, 0, 2, 3))).to(dev)
    yd = torch.from_numpy(ytr[:nb * bs].reshape(nb, bs)).to(dev)
    xv, yv = torch.from_numpy(Xva).to(dev), torch.from_numpy(yva).to(dev)
    schedule = [0] * args.epochs if not args.staged else [1] * args.epochs + [2] * args.epochs
    ckpt = args.ckpt or os.path.join(tempfile.mkdtemp(prefix="mfm_"), "mfn_%d.pt" % np.random.randint(0, 100000))
    c = cfg
    for epoch, stage in enumerate(schedule):
        if args.staged and epoch == args.epochs:
            best = 999999.0                                # the reference restarts the best-valid rule per stage (:348)
        model.train()
        acc = torch.zeros((), device=dev)
        for b in range(nb):
            losses = eng.train_step(Xd[b], yd[b], lr=lr, stage=stage)      # one C call: fwd + stage loss + bwd + Adam
            if stage == 0:
                acc += losses[0]                           # disc loss, accumulated on device (no per-step sync)
            else:
                gen = c["lda_xl"] * losses[1] + c["lda_xa"] * losses[2] + c["lda_xv"] * losses[3]
                acc += (gen if stage == 1 else losses[0]) + c["lda_mmd"] * losses[4]
        train_loss = acc.item() / nb
        # a hand-over inside a launch of the small-batch step can fail when something else runs on this GPU: the optimizer
        # skipped those steps, the engine switched to separate launches -- say so and carry on
        try:
            eng.check_status()
        except MfmError as err:
            print("note:", err)
        model.eval()
        out = eng.forward(xv, yv, train=False, want_xhat=False)
        valid_loss = eng.loss_dict(out["losses"])["disc"]
        if valid_loss < sched_best * (1 - 1e-4):
            sched_best, bad = valid_loss, 0
        else:
            bad += 1
            if bad > patience:
                lr, bad = lr * factor, 0
        if valid_loss <= best or args.staged:              # train_beta_vae saves every epoch (`if True:`, :343,353)
            best = valid_loss
            print(epoch, train_loss, valid_loss, "saving model")
            torch.save(model, ckpt)                        # the reference's checkpoint format: the whole module
        else:
            print(epoch, train_loss, valid_loss)
        sys.stdout.flush()
    model = torch.load(ckpt, weights_only=False)
    model.eval()
    with torch.no_grad():
        out = model.engine.forward(torch.from_numpy(Xte).to(dev), None, train=False, want_xhat=False)
    print("scoring y_hat")
    if ce:
        metrics.score_classes(out["y_hat"].cpu().numpy(), yte)
    else:
        metrics.score(out["y_hat"].squeeze(1).cpu().numpy(), yte)


if __name__ == "__main__":
    main()
