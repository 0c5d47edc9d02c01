"""hipGraph-captured controller step.

The controller is launch-bound by construction: Q*L*2 = 20 sequential LSTMCell steps on a batch of M = 6
(models/controller.py:73-145 in the reference) is ~200 micro-kernels for one sample() and, for the PPO update
(losses.py:132-151: 5 x {evaluate, backward, Adam}), ~5000 more -- ~35 ms of pure launch latency per
policy-search step on an MI355X, 30x the time of all augmentation / reward kernels together.  Both halves are
therefore captured once into two HIP graphs (torch.cuda.CUDAGraph == hipGraph on ROCm) and replayed:

  graph 1  sample:  policies, mean op/mag probs, sum log-prob, sum entropy   (no autograd; multinomial draws use
                    the graph-registered Philox generator state, so every replay draws fresh actions)
  graph 2  update:  PPO  -> 5 x { evaluate(policies) -> clipped surrogate -> backward -> Adam(capturable) }
                    REINFORCE -> teacher-forced rollout (same log-probs / entropies as the sample) -> loss ->
                    backward -> Adam

Same arithmetic, same parameter updates as the eager criterion objects in aadg_amd/losses.py (checked by
tests/test_gpu_controller_graph.py); the eager path remains the reference-API entry point and the CPU path.
"""
import torch

from ..losses import ProximalPolicyOptimization, Reinforce


class GraphedControllerStep(object):
    def __init__(self, controller, criterion, optimizer, M):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedControllerStep needs a GPU")
        self.controller, self.criterion, self.optimizer, self.M = controller, criterion, optimizer, M
        dev = next(controller.parameters()).device
        n_dec = controller.Q * controller.L * 2
        self.policies = torch.zeros(M, n_dec, dtype=torch.int64, device=dev)
        self.old_log_probs = torch.zeros(M, device=dev)
        self.reward = torch.zeros(M, device=dev)
        for group in optimizer.param_groups:
            group['capturable'] = True
        self._g_sample = self._g_update = None
        self._sample_out = self._update_out = None

    # ---- graph 1 -------------------------------------------------------------------------------------------
    def _sample_body(self):
        with torch.no_grad():
            policies, op_probs, mag_probs, log_probs, entropies = self.controller.sample(self.M)
            self.policies.copy_(policies)
            self.old_log_probs.copy_(log_probs)
        return op_probs, mag_probs, log_probs, entropies

    def sample(self):
        if self._g_sample is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._sample_body()
            torch.cuda.current_stream().wait_stream(side)
            self._g_sample = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_sample, capture_error_mode="thread_local"):
                self._sample_out = self._sample_body()
        self._g_sample.replay()
        op_probs, mag_probs, log_probs, entropies = self._sample_out
        return self.policies, op_probs, mag_probs, log_probs, entropies

    # ---- graph 2 -------------------------------------------------------------------------------------------
    def _update_body(self):
        crit, ctrl = self.criterion, self.controller
        if isinstance(crit, ProximalPolicyOptimization):
            total_loss = torch.zeros((), device=self.reward.device)
            for _ in range(crit.n_updates_per_iteration):
                ratios = torch.exp(ctrl.evaluate(self.policies, self.M) - self.old_log_probs)
                clipped = torch.clamp(ratios, 1 - crit.clip, 1 + crit.clip)
                loss = (-torch.min(ratios * self.reward, clipped * self.reward)).mean()
                self.optimizer.zero_grad(set_to_none=True)
                loss.backward()
                self.optimizer.step()
                total_loss = total_loss + loss.detach()
            mean_loss = total_loss / crit.n_updates_per_iteration
            return mean_loss, mean_loss
        if isinstance(crit, Reinforce):
            _, log_probs, entropies, _, _ = ctrl._rollout(self.M, forced=self.policies, want_entropy=True)
            log_probs = torch.stack(log_probs, dim=-1).sum(dim=-1)
            entropy_penalty = torch.stack(entropies, dim=-1).sum(dim=-1).mean()
            score_loss = (-log_probs * self.reward).mean()
            loss = score_loss - crit.penalty * entropy_penalty
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            self.optimizer.step()
            return loss.detach(), score_loss.detach()
        raise NotImplementedError(type(crit))

    def update(self, reward, entropies):
        """Same return contract as the criterion call: (loss, score_loss, entropy_penalty)."""
        self.reward.copy_(reward)
        if self._g_update is None:
            # warm-up iterations would move the parameters: snapshot and restore around them
            params = [p.detach().clone() for p in self.controller.parameters()]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._update_body()
            torch.cuda.current_stream().wait_stream(side)
            with torch.no_grad():
                for p, q in zip(self.controller.parameters(), params):
                    p.copy_(q)
            for state in self.optimizer.state.values():          # Adam moments / step back to zero
                for v in state.values():
                    if torch.is_tensor(v):
                        v.zero_()
            self.optimizer.zero_grad(set_to_none=True)
            self._g_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_update, capture_error_mode="thread_local"):
                self._update_out = self._update_body()
        self._g_update.replay()
        loss, score_loss = self._update_out
        return loss, score_loss, entropies.mean()


class FusedControllerStep(object):
    """Same interface as GraphedControllerStep, PPO only: the sample rollout and the 5 PPO epochs
    (evaluate -> clipped surrogate -> backward -> Adam) run as the fused HIP kernels of csrc/controller.hip
    (3 + 11 launches instead of ~1900 graph nodes).  Parameters and the optimizer's Adam state are updated in
    place, so `controller.state_dict()` / `optimizer.state_dict()` checkpoints stay the reference's."""

    fused = True                 # plain kernel launches (no graph capture): safe to call in the middle of a training step

    def __init__(self, controller, criterion, optimizer, M):
        from .. import _lib
        if not self.supported(controller, criterion, optimizer, M):
            raise RuntimeError("FusedControllerStep: unsupported controller / criterion / optimizer")
        self._lib = _lib
        self.controller, self.criterion, self.optimizer, self.M = controller, criterion, optimizer, M
        self.params = list(controller.parameters())
        self.ws = _lib.controller_workspace(controller, M)
        self.n_dec = controller.Q * controller.L * 2
        self.policies = self.old_log_probs = None
        group = optimizer.param_groups[0]
        self.lr, self.betas, self.eps = group['lr'], group['betas'], group['eps']
        for p in self.params:                                   # Adam state exactly as torch.optim.Adam lays it out
            state = optimizer.state[p]
            if len(state) == 0:
                state['step'] = torch.tensor(0.0, dtype=torch.float32)
                state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        self.exp_avg = [optimizer.state[p]['exp_avg'] for p in self.params]
        self.exp_avg_sq = [optimizer.state[p]['exp_avg_sq'] for p in self.params]
        self._dims = _lib.controller_dims(controller, M)
        self._ptrs = None                                       # pointer arrays of the calls: rebuilt when a parameter changed its storage

    def _pointers(self):
        p = self._ptrs
        if p is None or any(t.data_ptr() != q for t, q in zip(self.params, p[3])):
            p = self._ptrs = self._lib.controller_pointers(self.controller, self.exp_avg, self.exp_avg_sq)
        return p

    @staticmethod
    def supported(controller, criterion, optimizer, M):
        from .. import _lib
        if not (torch.cuda.is_available() and isinstance(criterion, ProximalPolicyOptimization)):
            return False
        if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
            return False
        g = optimizer.param_groups[0]
        plain = (not g.get('amsgrad', False) and g.get('weight_decay', 0) == 0 and not g.get('maximize', False) and
                 not g.get('capturable', False) and not g.get('fused', False) and not torch.is_tensor(g['lr']))
        same = len(g['params']) == 9 and all(a is b for a, b in zip(g['params'], controller.parameters()))
        on_host = all(not optimizer.state[p] or not optimizer.state[p]['step'].is_cuda for p in g['params'])
        return plain and same and on_host and _lib.controller_supported(controller, M)

    def sample(self):
        dev = self.params[0].device
        uniforms = torch.rand(self.M, self.n_dec, device=dev)
        policies, op_probs, mag_probs, log_probs, entropies = self._lib.controller_sample(self.controller, self.M, uniforms, self.ws,
                                                                                          self._pointers(), self._dims)
        self.policies, self.old_log_probs = policies, log_probs
        return policies, op_probs, mag_probs, log_probs, entropies

    def update(self, reward, entropies):
        """Same return contract as the criterion call: (loss, score_loss, entropy_penalty)."""
        crit = self.criterion
        group = self.optimizer.param_groups[0]                 # a scheduler may have moved the learning rate
        steps = self.optimizer.state[self.params[0]]['step']
        step0 = int(steps.item())
        n = crit.n_updates_per_iteration
        terms = self._lib.controller_ppo_update(self.controller, self.M, self.exp_avg, self.exp_avg_sq, self.policies,
                                                self.old_log_probs, reward.contiguous().float(), crit.clip, n, step0,
                                                group['lr'], group['betas'], group['eps'], self.ws, self._pointers(), self._dims)
        for p in self.params:
            self.optimizer.state[p]['step'] += n
        mean_loss = terms.mean()
        return mean_loss, mean_loss, entropies.mean()


def make_controller_step(controller, criterion, optimizer, M, fused=True):
    """Fastest available implementation of the controller's sample / update pair on the GPU."""
    if fused and FusedControllerStep.supported(controller, criterion, optimizer, M):
        return FusedControllerStep(controller, criterion, optimizer, M)
    return GraphedControllerStep(controller, criterion, optimizer, M)
