"""jt.nn subset: Module (execute), Parameter, functional ops, and Adam with Jittor's param_group internals."""
import torch as _torch
import torch.nn.functional as _F

functional = _F


def softmax(x, dim=None):
    return _F.softmax(x, dim=dim)


def bmm(a, b):
    return _torch.bmm(a, b)


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return _F.conv2d(x, weight, bias, stride, padding, dilation, groups)


conv = conv2d


def sign(x):
    return _torch.sign(x)


def relu(x):
    return _F.relu(x)


def Parameter(x, requires_grad=True):
    return x.detach().clone().requires_grad_(requires_grad)


class Module(_torch.nn.Module):
    """jittor.nn.Module: subclasses define execute(); calling the module runs it."""

    def forward(self, *a, **k):
        return self.execute(*a, **k)

    def execute(self, *a, **k):
        raise NotImplementedError


class _ParamList(list):
    """params of a group: whatever is stored becomes a leaf that requires grad, so a tensor the densifier builds
    (old[mask], concat(old, new), tensor.copy()) is trainable as soon as it is put back, as in Jittor."""

    @staticmethod
    def _leaf(t):
        if t is None:
            return t
        if t.is_leaf and t.requires_grad:
            return t
        if t.is_leaf:
            return t.requires_grad_(True)
        # A computed tensor cannot become a torch leaf: the group gets a trainable COPY.  Code that keeps using the object
        # it passed in (instead of re-reading param_groups[i]["params"][0], as the reference's densifier does) would
        # diverge from what is optimised, so say it once.
        import warnings
        warnings.warn("jittor compat: a non-leaf tensor was put into an optimizer group; the group holds a trainable copy - "
                      "re-read it from param_groups[...]['params']", RuntimeWarning, stacklevel=3)
        return t.detach().clone().requires_grad_(True)

    def __init__(self, it=()):
        super().__init__(self._leaf(t) for t in it)

    def append(self, t):
        super().append(self._leaf(t))

    def __setitem__(self, i, t):
        super().__setitem__(i, self._leaf(t))


class Adam:
    """jittor.nn.Adam(params | groups, lr, eps=1e-8, betas=(0.9, 0.999), weight_decay=0).
    param_groups[i] = {"params": [...], "grads": [...], "m": [...], "values": [...], "lr": ..., <user keys>};
    "values" is the second-moment estimate (Jittor's name).  step(loss) = backward(loss) + update."""

    def __init__(self, params, lr, eps=1e-8, betas=(0.9, 0.999), weight_decay=0):
        self.lr, self.eps, self.betas, self.weight_decay = lr, eps, betas, weight_decay
        self.n_step = 0
        self.param_groups = []
        params = list(params)
        if params and not isinstance(params[0], dict):
            params = [{"params": params}]
        for g in params:
            self.add_param_group(g)

    def add_param_group(self, group):
        g = dict(group)
        g["params"] = _ParamList(g["params"])
        g["grads"] = [None] * len(g["params"])
        g["m"] = [_torch.zeros_like(p) for p in g["params"]]
        g["values"] = [_torch.zeros_like(p) for p in g["params"]]
        self.param_groups.append(g)

    def zero_grad(self):
        for g in self.param_groups:
            g["grads"] = [None] * len(g["params"])

    def backward(self, loss, retain_graph=False):
        ps = [p for g in self.param_groups for p in g["params"]]
        gs = _torch.autograd.grad(loss, ps, retain_graph=retain_graph, allow_unused=True)
        it = iter(gs)
        for g in self.param_groups:
            new = []
            for k, p in enumerate(g["params"]):
                d = next(it)
                d = _torch.zeros_like(p) if d is None else d
                old = g["grads"][k] if k < len(g["grads"]) else None
                new.append(d if old is None else old + d)          # accumulates until zero_grad(), as Jittor does
            g["grads"] = new

    def pre_step(self, loss, retain_graph=False):
        if loss is not None:
            self.backward(loss, retain_graph)

    def step(self, loss=None, retain_graph=False):
        self.pre_step(loss, retain_graph)
        self.n_step += 1
        n = float(self.n_step)
        with _torch.no_grad():
            for g in self.param_groups:
                lr = g.get("lr", self.lr)
                eps = g.get("eps", self.eps)
                b0, b1 = g.get("betas", self.betas)
                wd = g.get("weight_decay", self.weight_decay)
                for k, p in enumerate(g["params"]):
                    d = g["grads"][k] if k < len(g["grads"]) else None
                    if d is None or p is None or p.numel() == 0:
                        continue
                    d = d + wd * p if wd else d
                    m, v = g["m"][k], g["values"][k]
                    m.mul_(b0).add_(d, alpha=1 - b0)
                    v.mul_(b1).addcmul_(d, d, value=1 - b1)
                    step_size = lr * (1 - b1 ** n) ** 0.5 / (1 - b0 ** n)
                    p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)
        if loss is not None:
            self.zero_grad()

    def state_dict(self):
        return {"n_step": self.n_step,
                "groups": [{k: ([t.detach().clone() if isinstance(t, _torch.Tensor) else t for t in v] if isinstance(v, list) else v)
                            for k, v in g.items() if k != "grads"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.n_step = sd.get("n_step", 0)
        for g, s in zip(self.param_groups, sd["groups"]):
            for k in ("m", "values"):
                g[k] = [t.detach().clone() for t in s[k]]
            with _torch.no_grad():
                for p, q in zip(g["params"], s["params"]):
                    p.data = q.detach().clone().to(p.device)
            for k, v in s.items():
                if k not in ("params", "m", "values"):
                    g[k] = v


class SGD(Adam):
    def step(self, loss=None, retain_graph=False):
        self.pre_step(loss, retain_graph)
        with _torch.no_grad():
            for g in self.param_groups:
                for k, p in enumerate(g["params"]):
                    d = g["grads"][k]
                    if d is not None:
                        p.add_(d, alpha=-g.get("lr", self.lr))
        if loss is not None:
            self.zero_grad()
