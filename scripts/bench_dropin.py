"""ms per step of the reference's UNCHANGED loop (mfm_mosi.py:427-441 incl. its per-step .item()) on MFM_KL_EF, B=32, T=20:
stock torch.optim.Adam vs factorized_amd.optim.Adam, per-tensor autograd path vs flat gradients, and the fused engine call."""
import os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs, synth
from factorized_amd.mfm_model import MFM_KL_EF
import factorized_amd.optim as optim

cfgs = configs.canonical_configs(dropout=True)
config = cfgs[0]
B, T = 32, 20
xn, yn = synth.make_batch(config["input_dims"], B, T, seed=7)
X, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
d_l, d_a, d_v = config["input_dims"]


def loop(model, optimizer, steps, item=True):
    criterion, gen_criterion = nn.L1Loss(), nn.MSELoss()
    epoch_loss = 0.0
    for _ in range(steps):
        optimizer.zero_grad()
        batch_X, batch_y = X, y
        decoded, mmd_loss, missing_loss = model.forward(batch_X)
        [x_l_hat, x_a_hat, x_v_hat, y_hat] = decoded
        gen_loss = config["lda_xl"] * gen_criterion(x_l_hat, batch_X[:, :, :d_l]) + config["lda_xa"] * gen_criterion(x_a_hat, batch_X[:, :, d_l:d_l + d_a]) \
            + config["lda_xv"] * gen_criterion(x_v_hat, batch_X[:, :, d_l + d_a:])
        disc_loss = criterion(y_hat.squeeze(1), batch_y)
        loss = disc_loss + gen_loss + config["lda_mmd"] * mmd_loss + missing_loss
        loss.backward()
        optimizer.step()
        if item:
            epoch_loss += disc_loss.item()


for name, opt_cls, fast, item in (("torch.optim.Adam, per-tensor autograd (round 2)", torch.optim.Adam, False, True),
                                  ("torch.optim.Adam, flat gradients", torch.optim.Adam, True, True),
                                  ("factorized_amd.optim.Adam, flat gradients", optim.Adam, True, True),
                                  ("factorized_amd.optim.Adam, flat gradients, no per-step .item()", optim.Adam, True, False)):
    # (the per-tensor runs leave cyclic garbage holding device tensors behind: measured, the next configuration's per-step
    # `.item()` then costs 0.2-0.3 ms more -- a property of this script's history, not of the path)
    import gc
    model = optimizer = None
    gc.collect(); torch.cuda.empty_cache()
    model = MFM_KL_EF(*cfgs)
    model.fast_grads = fast
    optimizer = opt_cls(model.parameters())
    model = model.to("cuda")
    model.train()
    loop(model, optimizer, 30, item)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(model, optimizer, 300, item)
    torch.cuda.synchronize()
    print("%-70s %.3f ms/step" % (name, 1e3 * (time.perf_counter() - t0) / 300))
model = MFM_KL_EF(*cfgs).to("cuda")
for _ in range(30):
    model.engine.train_step(X, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    model.engine.train_step(X, y)
torch.cuda.synchronize()
print("%-70s %.3f ms/step" % ("model.engine.train_step(X, y)  (one C call)", 1e3 * (time.perf_counter() - t0) / 300))

# eager, the other two classes of train_mfm (fused plan forward / backward since round 3 / 4)
from factorized_amd import mfm_model as M2
for cls_name in ("MFM_KL", "MFM"):
    model = getattr(M2, cls_name)(*cfgs)
    optimizer = optim.Adam(model.parameters())
    model = model.to("cuda")
    model.train()
    loop(model, optimizer, 30, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(model, optimizer, 300, True)
    torch.cuda.synchronize()
    print("%-70s %.3f ms/step" % ("factorized_amd.optim.Adam, flat gradients, %s" % cls_name, 1e3 * (time.perf_counter() - t0) / 300))

# the same loop captured once into a hipGraph (train.GraphedModuleStep: fused plan with device-side epochs / dropout streams,
# optim.Adam(capturable=True)) and replayed; per-step input copy into the static batch included
from factorized_amd import train
for cls_name in ("MFM_KL_EF", "MFM_KL", "MFM"):
    from factorized_amd import mfm_model as M
    model = getattr(M, cls_name)(*cfgs).to("cuda")
    model.train()
    gs = train.GraphedModuleStep(model, config, B, T, lr=1e-3)
    for _ in range(30):
        gs.step(X, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        gs.step(X, y)
    torch.cuda.synchronize()
    print("%-70s %.3f ms/step" % ("GraphedModuleStep(%s): the unchanged loop as one hipGraph replay" % cls_name, 1e3 * (time.perf_counter() - t0) / 300))
    assert float(gs.loss) == float(gs.loss) and model.engine.check_status() == 0

if os.environ.get("MFM_DROPIN_PROFILE"):
    import cProfile, pstats
    model = MFM_KL_EF(*cfgs)
    optimizer = optim.Adam(model.parameters())
    model = model.to("cuda"); model.train()
    loop(model, optimizer, 30, False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    loop(model, optimizer, 300, False)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)

if os.environ.get("MFM_DROPIN_SECTIONS"):
    model = MFM_KL_EF(*cfgs)
    optimizer = optim.Adam(model.parameters())
    model = model.to("cuda"); model.train()
    criterion, gen_criterion = nn.L1Loss(), nn.MSELoss()
    acc = [0.0] * 5
    pc = time.perf_counter
    for it in range(330):
        t0 = pc()
        optimizer.zero_grad()
        t1 = pc()
        decoded, mmd_loss, missing_loss = model.forward(X)
        t2 = pc()
        [x_l_hat, x_a_hat, x_v_hat, y_hat] = decoded
        gen_loss = config["lda_xl"] * gen_criterion(x_l_hat, X[:, :, :d_l]) + config["lda_xa"] * gen_criterion(x_a_hat, X[:, :, d_l:d_l + d_a]) \
            + config["lda_xv"] * gen_criterion(x_v_hat, X[:, :, d_l + d_a:])
        disc_loss = criterion(y_hat.squeeze(1), y)
        loss = disc_loss + gen_loss + config["lda_mmd"] * mmd_loss + missing_loss
        t3 = pc()
        loss.backward()
        t4 = pc()
        optimizer.step()
        t5 = pc()
        if it >= 30:
            for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[k] += d
        if it % 50 == 49:
            torch.cuda.synchronize()
    print("host us/step: zero_grad %.0f  forward %.0f  loss ops %.0f  backward %.0f  step %.0f  (sum %.0f)" %
          tuple([1e6 * a / 300 for a in acc] + [1e6 * sum(acc) / 300]))
