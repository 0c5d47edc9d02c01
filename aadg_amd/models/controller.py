"""Policy controller -- API mirror of the reference's models/controller.py:9-145.

Same parameters (state_dict keys `embedding`, `lstm`, `outop`, `outmag`, so a reference
`final_controller_state.pth` loads), same initialisation (U(-0.1, 0.1), head biases 0), same sampling
law: Q sub-policies x L (op, magnitude) pairs, LSTM state reset per sub-policy, logits squashed as
C*tanh(z)/T, op tokens embedded at [0, NUM_OPS) and magnitude tokens at NUM_OPS + m.
`forward(M)` / `sample(M)` return (policies [M, Q*L*2] int64, mean op probs, mean mag probs,
sum log-prob [M], sum entropy [M]); `evaluate(policies, M)` returns the teacher-forced sum log-prob.

Unlike the reference (hard-wired `.cuda()`, models/controller.py:42-53) the module runs on whatever
device its parameters live on.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..data.basic import augment_list


class Controller(nn.Module):
    def __init__(self, cfg, n_subpolicies=5, embedding_dim=32, hidden_dim=100):
        super(Controller, self).__init__()
        self.L = cfg.CONTROLLER.L
        self.T = cfg.CONTROLLER.T
        self.C = cfg.CONTROLLER.C
        self.NUM_MAGS = cfg.CONTROLLER.NUM_MAGS
        n_excluded = len(cfg.CONTROLLER.EXCLUDE_OPS) or cfg.CONTROLLER.EXCLUDE_OPS_NUM
        self.NUM_OPS = len(augment_list()) - n_excluded
        self.Q = n_subpolicies
        self.embedding_dim = embedding_dim
        self.hidden_dim = hidden_dim

        self.embedding = nn.Embedding(self.NUM_OPS + self.NUM_MAGS, embedding_dim)
        self.lstm = nn.LSTMCell(embedding_dim, hidden_dim)
        self.outop = nn.Linear(hidden_dim, self.NUM_OPS)
        self.outmag = nn.Linear(hidden_dim, self.NUM_MAGS)
        # token offset of each of the 2L decisions of a sub-policy (op tokens at 0, magnitude tokens at NUM_OPS);
        # a non-persistent buffer: not in the state_dict, but resident on the device (hipGraph capture forbids H2D)
        self.register_buffer("_tok_offsets", torch.tensor([0, self.NUM_OPS] * self.L, dtype=torch.long), persistent=False)
        self.init_parameters()

    def init_parameters(self):
        for param in self.parameters():
            param.data.uniform_(-0.1, 0.1)
        self.outop.bias.data.fill_(0)
        self.outmag.bias.data.fill_(0)

    # -- helpers ---------------------------------------------------------------------------------
    def _fresh_state(self, batch_size):
        ref = self.embedding.weight
        zeros = lambda width: torch.zeros(batch_size, width, dtype=ref.dtype, device=ref.device)  # noqa: E731
        return zeros(self.embedding_dim), zeros(self.hidden_dim), zeros(self.hidden_dim)

    def _log_policy(self, logits):
        return F.log_softmax(self.C * torch.tanh(logits) / self.T, dim=-1)

    def _decisions(self):
        """Order of the 2*L*Q decisions: (head, token offset) per step."""
        for _ in range(self.L):
            yield self.outop, 0
            yield self.outmag, self.NUM_OPS

    def _rollout(self, batch_size, forced=None, want_entropy=False):
        """Shared by sample (forced=None: draw actions) and evaluate (forced = policies: teacher forcing)."""
        actions, log_probs, entropies, probs_op, probs_mag = [], [], [], [], []
        col = 0
        for _ in range(self.Q):
            inp, hx, cx = self._fresh_state(batch_size)
            for head, offset in self._decisions():
                hx, cx = self.lstm(inp, (hx, cx))
                logits = head(hx)
                logp = self._log_policy(logits)
                if forced is None or want_entropy:
                    p = F.softmax(self.C * torch.tanh(logits) / self.T, dim=-1)
                    entropies.append(-(logp * p).sum(1))
                    (probs_op if offset == 0 else probs_mag).append(p)
                if forced is None:
                    act = p.multinomial(num_samples=1)[:, 0]
                else:
                    act = forced[:, col].long()
                actions.append(act)
                log_probs.append(logp.gather(1, act.unsqueeze(-1))[:, 0])
                inp = self.embedding(offset + act)
                col += 1
        return actions, log_probs, entropies, probs_op, probs_mag

    # -- reference API ---------------------------------------------------------------------------
    def forward(self, batch_size=1):
        return self.sample(batch_size)

    def sample(self, batch_size=1):
        actions, log_probs, entropies, probs_op, probs_mag = self._rollout(batch_size)
        policies = torch.stack(actions, dim=-1)
        op_probs = torch.stack(probs_op, dim=-1).permute(0, 2, 1).reshape(-1, self.NUM_OPS)
        mag_probs = torch.stack(probs_mag, dim=-1).permute(0, 2, 1).reshape(-1, self.NUM_MAGS)
        return (policies, op_probs.mean(dim=0), mag_probs.mean(dim=0),
                torch.stack(log_probs, dim=-1).sum(dim=-1), torch.stack(entropies, dim=-1).sum(dim=-1))

    def evaluate(self, policies, batch_size):
        """Teacher-forced sum of log-probabilities of `policies` [M, Q*L*2] (models/controller.py:118-145).

        Same arithmetic as replaying the 2*L*Q LSTMCell steps one by one, restructured for the GPU: with the
        actions given, the Q sub-policies are independent sequences (the reference resets the state per
        sub-policy), so they run as ONE batch of M*Q sequences of length 2*L, the input-to-hidden product is a
        single GEMM over all steps, and the two heads / log-softmax / gather run once over all steps.  That is
        ~25 kernels instead of ~240 per call (the PPO update calls this 5 times, forward and backward)."""
        M, Q, S, H = batch_size, self.Q, 2 * self.L, self.hidden_dim
        acts = policies.reshape(M, Q, S).long()
        tokens = acts + self._tok_offsets                                             # op tokens / NUM_OPS + mag tokens
        emb = self.embedding(tokens[:, :, :-1])                              # inputs of steps 1..S-1
        x = torch.cat([emb.new_zeros(M, Q, 1, self.embedding_dim), emb], dim=2).reshape(M * Q, S, self.embedding_dim)
        gi = F.linear(x, self.lstm.weight_ih, self.lstm.bias_ih + self.lstm.bias_hh)   # [MQ, S, 4H], gate order i,f,g,o
        h = gi.new_zeros(M * Q, H)
        c = gi.new_zeros(M * Q, H)
        hs = []
        for t in range(S):
            gates = gi[:, t] + F.linear(h, self.lstm.weight_hh)
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs.append(h)
        hs = torch.stack(hs, dim=1)                                          # [MQ, S, H]
        a = acts.reshape(M * Q, S)
        lp_op = self._log_policy(self.outop(hs[:, 0::2])).gather(2, a[:, 0::2].unsqueeze(-1))
        lp_mag = self._log_policy(self.outmag(hs[:, 1::2])).gather(2, a[:, 1::2].unsqueeze(-1))
        return (lp_op.sum(dim=(1, 2)) + lp_mag.sum(dim=(1, 2))).reshape(M, Q).sum(dim=1)

    def evaluate_stepwise(self, policies, batch_size):
        """The reference's step-by-step formulation (kept for tests: `evaluate` must agree with it)."""
        _, log_probs, _, _, _ = self._rollout(batch_size, forced=policies)
        return torch.stack(log_probs, dim=-1).sum(dim=-1)
