#!/usr/bin/env python3
"""Python-3 restatement of the reference's MOSI driver loop on the MI355X path.

The reference README tells users to run `mfm_test_mosi.py --config configs/mosi.json`
(README.md:36); the nearest real file is the Python-2 `mfm_mosi.py`.  This script keeps its
shape -- argparse `--config`, `seqlength` read from the JSON (mfm_mosi.py:33-47), one shuffle with
numpy seed 123 (:387-389), floor-division batch count (:423), Adam defaults (:403),
ReduceLROnPlateau('min') on the validation L1 (:417,472), best-checkpoint rule (:473-477), the
`epoch train valid` log line (:476-479) and score() metrics (:483-499) -- but trains on synthetic
MOSI-shape data (the CMU-MOSI pickles are private) with the fused MI355X step.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, synth  # noqa: E402
from factorized_amd.mfm_model import MFM, MFM_KL, MFM_KL_EF  # noqa: E402
from factorized_amd.train import GraphedModuleStep  # noqa: E402


def score(pred, y):
    mae = float(np.mean(np.absolute(pred - y)))
    corr = float(np.corrcoef(pred, y)[0][1])
    mult = round(float(np.sum(np.round(pred) == np.round(y))) / float(len(y)), 5)
    acc = float(np.mean((pred >= 0) == (y >= 0)))
    print("mae: ", mae); print("corr: ", corr); print("mult_acc: ", mult); print("Accuracy ", acc)
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(os.path.dirname(__file__), "..", "configs", "mosi.json"))
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--n-train", type=int, default=1280)
    ap.add_argument("--model", choices=["kl_ef", "kl", "mmd"], default="kl_ef",
                    help="kl_ef: MFM_KL_EF on the fused one-call engine; kl / mmd: MFM_KL / MFM, the classes "
                         "train_mfm picks by config['type'] (mfm_mosi.py:398-401), on the module path with the "
                         "reference-style step replayed as a hipGraph")
    ap.add_argument("--staged", action="store_true",
                    help="train_beta_vae schedule (mfm_mosi.py:225-361): `epochs` of stage 1 (gen + reg loss), then "
                         "`epochs` of stage 2 (disc + reg loss), instead of the joint loss of train_mfm")
    args = ap.parse_args()
    _, T = C.load_json_config(args.config)                  # only `seqlength` is read, as in the reference
    cfgs = C.canonical_configs(dropout=True)
    cfg = cfgs[0]
    np.random.seed(123)
    Xtr, ytr = synth.make_dataset(cfg["input_dims"], args.n_train, T, seed=11)
    Xva, yva = synth.make_dataset(cfg["input_dims"], 229, T, seed=12)
    Xte, yte = synth.make_dataset(cfg["input_dims"], 686, T, seed=13)
    p = np.random.permutation(Xtr.shape[1])
    Xtr, ytr = Xtr[:, p], ytr[p]
    dev = torch.device("cuda")
    if args.model != "kl_ef":
        if args.staged:
            raise SystemExit("--staged is implemented on the fused MFM_KL_EF engine only")
        return train_module_path(args, cfgs, T, (Xtr, ytr), (Xva, yva), (Xte, yte), dev)
    model = MFM_KL_EF(*cfgs).to(dev)
    eng = model.engine                                     # fused step on the module's own storage
    lr = 1e-3
    best, bad, factor, patience = 999999.0, 0, 0.1, 10     # ReduceLROnPlateau('min') defaults
    bs = cfg["batchsize"]
    nb = Xtr.shape[1] // bs
    Xd = torch.from_numpy(np.ascontiguousarray(Xtr[:, :nb * bs].reshape(T, nb, bs, -1).transpose(1, 0, 2, 3))).to(dev)
    yd = torch.from_numpy(ytr[:nb * bs].reshape(nb, bs)).to(dev)
    xv, yv = torch.from_numpy(Xva).to(dev), torch.from_numpy(yva).to(dev)
    schedule = [0] * args.epochs if not args.staged else [1] * args.epochs + [2] * args.epochs
    c = cfg
    for epoch, stage in enumerate(schedule):
        if args.staged and epoch == args.epochs:
            best = 999999.0                                # the reference restarts the best-valid rule per stage (:348)
        model.train()
        acc = torch.zeros((), device=dev)
        for b in range(nb):
            if stage == 0:
                losses = eng.train_step(Xd[b], yd[b], lr=lr)
                acc += losses[0]                           # disc loss, accumulated on device (no per-step sync)
            else:
                losses = eng.forward(Xd[b], yd[b], train=True, want_xhat=False)["losses"]
                eng.backward(Xd[b], yd[b], stage=stage)    # gradients of gen+reg (1) or disc+reg (2), mfm_mosi.py:278-281
                eng.adam(lr=lr)
                gen = c["lda_xl"] * losses[1] + c["lda_xa"] * losses[2] + c["lda_xv"] * losses[3]
                acc += (gen if stage == 1 else losses[0]) + c["lda_mmd"] * losses[4]
        train_loss = acc.item() / nb
        model.eval()
        out = eng.forward(xv, yv, train=False, want_xhat=False)
        valid_loss = eng.loss_dict(out["losses"])["disc"]
        if valid_loss < best * (1 - 1e-4):
            bad = 0
        else:
            bad += 1
            if bad > patience:
                lr, bad = lr * factor, 0
        if valid_loss <= best:
            best = valid_loss
            print(epoch, train_loss, valid_loss, "saving model")
            best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        else:
            print(epoch, train_loss, valid_loss)
        sys.stdout.flush()
    model.load_state_dict(best_state)
    model.eval()
    with torch.no_grad():
        decoded, _, _ = model.forward(torch.from_numpy(Xte).to(dev))
    print("scoring y_hat")
    score(decoded[3].squeeze(1).cpu().numpy(), yte)


def train_module_path(args, cfgs, T, train_set, valid_set, test_set, dev):
    """train_mfm (mfm_mosi.py:386-503) for MFM_KL / MFM: same loop, the step body replayed as a hipGraph."""
    cfg = cfgs[0]
    (Xtr, ytr), (Xva, yva), (Xte, yte) = train_set, valid_set, test_set
    model = (MFM_KL if args.model == "kl" else MFM)(*cfgs).to(dev)
    model.train()
    bs = cfg["batchsize"]
    nb = Xtr.shape[1] // bs                                  # floor division, tail dropped (:423)
    Xd = torch.from_numpy(np.ascontiguousarray(Xtr[:, :nb * bs].reshape(T, nb, bs, -1).transpose(1, 0, 2, 3))).to(dev)
    yd = torch.from_numpy(ytr[:nb * bs].reshape(nb, bs)).to(dev)
    xv, yv = torch.from_numpy(Xva).to(dev), torch.from_numpy(yva).to(dev)
    stepper = GraphedModuleStep(model, cfg, bs, T, lr=1e-3)
    l1 = torch.nn.L1Loss()
    lr, best, bad, factor, patience = 1e-3, 999999.0, 0, 0.1, 10     # ReduceLROnPlateau('min') defaults (:417)
    best_state = None
    for epoch in range(args.epochs):
        model.train()
        acc = torch.zeros((), device=dev)
        for b in range(nb):
            _, disc = stepper.step(Xd[b], yd[b])
            acc += disc                                      # accumulated on the device: no per-step .item()
        train_loss = acc.item() / nb
        model.eval()
        with torch.no_grad():
            decoded, _, _ = model.forward(xv)
            valid_loss = l1(decoded[3].squeeze(1), yv).item()
        if valid_loss < best * (1 - 1e-4):
            bad = 0
        else:
            bad += 1
            if bad > patience:
                lr, bad = lr * factor, 0
                stepper.set_lr(lr)
        if valid_loss <= best:
            best = valid_loss
            print(epoch, train_loss, valid_loss, "saving model")
            best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        else:
            print(epoch, train_loss, valid_loss)
        sys.stdout.flush()
    model.load_state_dict(best_state)
    model.eval()
    with torch.no_grad():
        decoded, _, _ = model.forward(torch.from_numpy(Xte).to(dev))
    print("scoring y_hat")
    score(decoded[3].squeeze(1).cpu().numpy(), yte)


if __name__ == "__main__":
    main()
