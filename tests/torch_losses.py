"""Test infrastructure: the four loss items of Trainer.backward (models/trainer.py:154-178) restated with torch ops on
the device, and a training iteration that feeds their autograd cotangents into the native backward.  The cross-check of
the native loss kernels (ffr_train_losses); never imported by the product package."""
import torch
import torch.nn.functional as F

from ffrnet_amd.train import average_gradients

TRIPLET_MARGIN = 0.1     # models/trainer.py:39


def cosine_sim(x1, x2):
    """models/recnet.py:220-224."""
    return torch.bmm(F.normalize(x1, dim=2), F.normalize(x2, dim=2).permute(0, 2, 1))


def self_similarity(x):
    """selfSimilarity, models/recnet.py:226-236 -> (ss_space [N,HW,H,W], ss_channel [N,C,C])."""
    n, c, h, w = x.shape
    v = x.reshape(n, c, h * w)
    vt = v.permute(0, 2, 1)
    return cosine_sim(vt, vt).reshape(n, h * w, h, w), cosine_sim(v, v)


def triplet_loss(x, y, z):
    """TripletLoss.forward, models/trainer.py:38-43."""
    pos = 1 - torch.sum(F.normalize(x) * F.normalize(y), 1)
    neg = 1 - torch.sum(F.normalize(x) * F.normalize(z), 1)
    return F.relu((pos - neg) + TRIPLET_MARGIN).mean(), pos.mean(), neg.mean()


def trainer_losses(f_non, f_ocl, pred_loss_non, pred_loss_ocl, space_non, space_ocl, channel_non, channel_ocl,
                   feat_map_non, f_enc_non, f_enc_ocl, label, loss_weight=(1, 1, 1, 1)):
    """The four weighted loss items of Trainer.backward, models/trainer.py:154-178."""
    ss_space, ss_channel = self_similarity(feat_map_non)
    ss_space_non, _ = self_similarity(space_non)
    ss_space_ocl, _ = self_similarity(space_ocl)
    _, ss_channel_non = self_similarity(channel_non)
    _, ss_channel_ocl = self_similarity(channel_ocl)
    mse = F.mse_loss
    l_space = (mse(ss_space, ss_space_non) + mse(ss_space, ss_space_ocl)) / 2
    l_channel = (mse(ss_channel, ss_channel_non) + mse(ss_channel, ss_channel_ocl)) / 2
    items = [(l_space + l_channel) / 2,
             triplet_loss(f_ocl, f_enc_non, f_enc_ocl)[0],
             (mse(f_non, f_enc_non) + mse(f_ocl, f_enc_non)) / 2,
             F.cross_entropy(pred_loss_non, label) / (1e-8 + loss_weight[3]) + F.cross_entropy(pred_loss_ocl, label)]
    return [l * w for l, w in zip(items, loss_weight)]



def step_torch_losses(trainer, img_non, img_ocl, label):
    """NativeTrainer.step with the four loss items evaluated by torch ops (`trainer_losses`) and their autograd
    cotangents handed to ffr_train_backward."""
    eng = trainer.engine
    n = img_non.size(0)
    with torch.no_grad():
        fm, f_enc = eng.encoder_forward(torch.cat((img_non, img_ocl), 0))
    label = label.to(fm.device)
    outs = eng.train_forward(fm, torch.cat((label, label)), groups=2,
                             want=('f_new', 'pred_loss', 'pred_label', 'feat_space', 'feat_channel'))
    f_new, pred_loss, pred_label, _, _, feat_space, feat_channel = outs
    leaves = [t.detach().requires_grad_(True) for t in (f_new, pred_loss, feat_space, feat_channel)]
    lf, lp, ls, lc = leaves
    items = trainer_losses(lf[:n], lf[n:], lp[:n], lp[n:], ls[:n], ls[n:], lc[:n], lc[n:], fm[:n], f_enc[:n],
                           f_enc[n:], label.long(), trainer.loss_weight)
    torch.autograd.backward(sum(items))
    eng.train_zero_grad()
    eng.train_backward([lf.grad, lp.grad, None, None, None, ls.grad, lc.grad])
    average_gradients(trainer.flat_grads, trainer.group)
    eng.train_adam_step(trainer.lr, trainer.betas, 1e-8, trainer.weight_decay, trainer.clip_value)
    trainer.accuracy = (pred_label[n:].argmax(1) == label).float().mean()
    trainer.loss_items = [l.detach() for l in items]
    return trainer.loss_items
