"""The three WGAN-GP minibatch steps on CPU (oracle; test infrastructure).

Restates ``train.py:18-104`` (critic_x_iteration), ``train.py:107-186``
(critic_z_iteration) and ``train.py:189-249`` (decoder_iteration) for CPU
tensors.  The host-RNG draws keep the reference's order and sources (NumPy's
global generator for z, torch's CPU generator for alpha: SURVEY.md D9) so that a
seeded run consumes exactly the same random numbers; ``z=`` / ``alpha=`` inject
them instead.  Reference quirks kept on purpose: the gradient-penalty norm is
taken over the whole ``(1, B*S)`` batch (D8), and ``critic_x_iteration`` runs its
penalty in float64 when the sample is float64.
"""
import numpy as np
import torch
from torch.autograd import grad as torch_grad

from . import gmath


def _draw_z(params, z):
    if z is None:
        z = np.random.normal(size=(1, params.batch_size, params.latent_space_dim))  # train.py:24
    return torch.as_tensor(np.asarray(z), dtype=torch.float32).reshape(1, params.batch_size, -1)


def _draw_alpha(shape, alpha):
    if alpha is None:
        return torch.rand(shape)                                                     # train.py:64,149
    return torch.as_tensor(alpha, dtype=torch.float32).reshape(shape)


def _gradient_penalty(critic, real, fake, alpha):
    """train.py:58-93 / 143-178."""
    inter = (alpha * real.detach() + (1 - alpha) * fake.detach()).requires_grad_(True)
    prob = critic(inter)
    g = torch_grad(outputs=prob, inputs=inter, grad_outputs=torch.ones(prob.size()),
                   create_graph=True, retain_graph=True)[0]
    g = g.view(real.size(0), -1)                       # real.size(0) == 1: one norm for the whole batch
    return ((torch.sqrt(torch.sum(g ** 2, dim=1) + 1e-12) - 1) ** 2).mean()


def critic_x_iteration(sample, decoder, critic_x, optim_cx, params, z=None, alpha=None):
    optim_cx.zero_grad()
    y = sample.view(1, params.batch_size, params.signal_shape)
    valid = torch.squeeze(critic_x(y))
    z = _draw_z(params, z)
    x_ = decoder(z)[0] if decoder.hyperbolic else decoder(z)                          # train.py:27-33
    fake = torch.squeeze(critic_x(x_))
    wl = torch.mean(fake) + torch.mean(-valid)                                        # train.py:37-42,98
    gp = _gradient_penalty(critic_x, y, x_, _draw_alpha(y.shape, alpha))
    loss = wl + 10 * gp
    loss.backward(retain_graph=True)
    optim_cx.step()
    return loss


def critic_z_iteration(sample, encoder, critic_z, optim_cz, params, z=None, alpha=None):
    optim_cz.zero_grad()
    x = sample.view(1, params.batch_size, params.signal_shape)
    z_ = encoder(x)
    fake = torch.squeeze(critic_z(z_))
    z = _draw_z(params, z)
    valid = torch.squeeze(critic_z(z))
    wl = torch.mean(fake) + torch.mean(-valid)                                        # train.py:113-125
    gp = _gradient_penalty(critic_z, z, z_, _draw_alpha(z.shape, alpha))
    loss = wl + 10 * gp
    loss.backward(retain_graph=True)
    optim_cz.step()
    return loss


def decoder_iteration(sample, encoder, decoder, critic_x, critic_z, optim_dec, params, z=None):
    optim_dec.zero_grad()
    x = sample.view(1, params.batch_size, params.signal_shape)
    z_enc = encoder(x)
    fake_z = critic_z(z_enc)
    z = _draw_z(params, z)
    x_gen = decoder(z)[0] if decoder.hyperbolic else decoder(z)
    fake_x = critic_x(x_gen)
    adv = torch.mean(-fake_x) + torch.mean(-fake_z)                                   # train.py:212-217
    if decoder.hyperbolic:
        x_rec, _ = decoder(z_enc)
        hyper_x = decoder.hyperbolic_linear(x.view(-1, params.signal_shape))
        dist = gmath.rowwise_poincare_distance(x_rec, hyper_x)                        # train.py:226-230
        hyper_loss = torch.div(torch.sum(dist), params.batch_size)
        loss = 10 * hyper_loss + adv
        loss.backward(retain_graph=True)
        optim_dec.step()
        return loss, hyper_loss, torch.Tensor([0])
    x_rec = decoder(z_enc)
    mse = torch.nn.functional.mse_loss(x_rec.float(), x.float())                      # train.py:241-242
    loss = 10 * mse + adv
    loss.backward()
    optim_dec.step()
    return loss, 0, mse


def set_trainable(modules, flag):
    """train.py:306-313 / 333-340."""
    for m in modules:
        for p in m.parameters():
            p.requires_grad = flag


def make_optimizers(encoder, decoder, critic_x, critic_z, params):
    """train.py:274-288."""
    from .radam import RiemannianAdam
    ocx = torch.optim.Adam(critic_x.parameters(), lr=params.lr, betas=(0.9, 0.999))
    ocz = torch.optim.Adam(critic_z.parameters(), lr=params.lr, betas=(0.9, 0.999))
    gen = list(decoder.parameters()) + list(encoder.parameters())
    if params.hyperbolic:
        odec = RiemannianAdam(gen, lr=params.lr, weight_decay=1e-5, stabilize=10)
    else:
        odec = torch.optim.Adam(gen, lr=params.lr, betas=(0.9, 0.999))
    return ocx, ocz, odec


def train_epoch(batches, encoder, decoder, critic_x, critic_z, optims, params, n_critics=5):
    """One epoch of train.py:299-356 over an in-memory list of (B,S,1) float64 batches."""
    ocx, ocz, odec = optims
    set_trainable((decoder, encoder), False)
    set_trainable((critic_x, critic_z), True)
    for _ in range(n_critics):
        for s in batches:
            critic_x_iteration(s, decoder, critic_x, ocx, params)
            critic_z_iteration(s, encoder, critic_z, ocz, params)
    set_trainable((decoder, encoder), True)
    set_trainable((critic_x, critic_z), False)
    out = None
    for s in batches:
        out = decoder_iteration(s, encoder, decoder, critic_x, critic_z, odec, params)
    return out
