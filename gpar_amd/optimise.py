"""L-BFGS-B driver over a `Vars` store (counterpart of `varz.torch.minimise_l_bfgs_b`, which the reference calls
at /root/reference/gpar/regression.py:459).

The objective receives the variable store and returns a torch scalar; its gradient with respect to the selected
latent (unconstrained) variables is obtained by back-propagation — for the GP layers that is the analytic
gradient computed on the GPU (gpar_amd/gp.py: `_LogMarginal`), chained through the bound transforms by torch.
"""
import logging

import numpy as np
import scipy.optimize
import torch

from .engine import NotPositiveDefiniteError

__all__ = ["minimise_l_bfgs_b", "evaluation_count"]

log = logging.getLogger(__name__)

_evaluations = 0  # objective + gradient evaluations made by this process (read by bench.py to report work done)


def evaluation_count():
    return _evaluations


def count_evaluation():
    """One more objective + gradient evaluation (several host threads may train layers at once: a statistic, not a synchronised counter)."""
    global _evaluations
    _evaluations += 1


def minimise_l_bfgs_b(f, vs, names=None, iters=1000, f_calls=10000, trace=False):
    """Minimise `f(vs)` over the variables whose names match `names` (globs allowed; default: all).

    Returns the final objective value; the optimum is written back into `vs`.
    """
    fg, names, x0 = objective_and_gradient(f, vs, names, trace=trace)
    x_opt, val, _ = scipy.optimize.fmin_l_bfgs_b(fg, x0, maxiter=iters, maxfun=f_calls)
    vs.set_vector(x_opt, names)
    return val


def objective_and_gradient(f, vs, names=None, trace=False):
    """(fg, names, x0): `fg(x)` -> (value, gradient) of `f(vs)` at the latent vector x of the variables matching `names` - the
    function scipy's L-BFGS-B calls - by back-propagation; the resolved names; the current latent vector."""
    patterns = names

    def select():
        return vs.match(patterns) if patterns is not None else vs.names

    names = select()
    unmatched = patterns is not None and any(not vs.match(p) for p in ([patterns] if isinstance(patterns, str) else patterns))
    if not names or unmatched:
        # The objective creates its variables lazily (reference regression.py:92-180), and varz evaluates it once before it
        # resolves the names: with fix=False the patterns "0/*" .. "{pi}/*" must pick up layer pi's variables, which do not
        # exist before the first evaluation of the (pi + 1)-layer model.  Evaluated once here whenever a pattern matches nothing.
        with torch.no_grad():
            f(vs)
        names = select()
    latents = vs.get_vars(*names)
    if not latents:
        raise ValueError("no variables to optimise")
    x0 = vs.get_vector(names)

    def fg(x):
        count_evaluation()
        vs.set_vector(x, names)
        previous = [t.requires_grad for t in latents]
        for t in latents:
            t.requires_grad_(True)
            t.grad = None
        try:
            value = f(vs)
            value.backward()
            grad = np.concatenate(
                [(t.grad if t.grad is not None else torch.zeros_like(t)).detach().numpy().reshape(-1) for t in latents]
            )
            val = float(value.detach())
        except (NotPositiveDefiniteError, ArithmeticError) as e:  # as varz: report NaN and let the line search back off
            log.warning("objective evaluation failed (%s); returning NaN", e)
            val, grad = np.nan, np.zeros_like(x)
        finally:
            for t, r in zip(latents, previous):
                t.requires_grad_(r)
                t.grad = None
        if trace:
            print(f"  objective {val:.6e}  |grad| {np.linalg.norm(grad):.3e}")
        return val, grad

    return fg, names, x0
