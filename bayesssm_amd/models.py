"""Built-in device models: the stand-ins for the reference's R closures.

The reference takes init_fn / transition_fn / log_likelihood_fn as arbitrary R
closures (R/particle_filter-doc.R:7-35).  Closures cannot run on a GPU, so the
device path takes DESCRIPTORS of built-in models with the same argument names;
model parameters still travel as named arguments (`phi=..., sigma_x=...`), as
they do through `...` in the reference.
"""


class ModelFn:
    """One of the three (four, with the APF) user functions of a built-in model."""

    def __init__(self, model, role, params):
        self.model = model          # "lg" | "ar1sin"
        self.role = role            # "init" | "transition" | "log_likelihood" | "aux_log_likelihood"
        self.params = tuple(params)  # model-specific named arguments this function reads

    def formals(self):
        """Argument names as the reference's .check_params_match sees them (R/utils.R:19-61)."""
        head = {"init": ["num_particles"], "transition": ["particles"],
                "log_likelihood": ["y", "particles"], "aux_log_likelihood": ["y", "particles"]}[self.role]
        return head + list(self.params)

    def __repr__(self):
        return "<%s %s_fn(%s)>" % (self.model, self.role, ", ".join(self.formals()))


class Model:
    PARAM_ORDER = ("phi", "sigma_x", "sigma_y")

    def __init__(self, name, doc):
        self.name = name
        self.__doc__ = doc
        self.init_fn = ModelFn(name, "init", ())
        self.transition_fn = ModelFn(name, "transition", ("phi", "sigma_x"))
        self.log_likelihood_fn = ModelFn(name, "log_likelihood", ("sigma_y",))
        self.aux_log_likelihood_fn = ModelFn(name, "aux_log_likelihood", ("phi", "sigma_y"))
        self.dim = 1
        self.param_order = self.PARAM_ORDER
        self.constants = ()

    def rw_move_fn(self, sd=0.1):
        return MoveFn(self.name, sd)


class SirModel(Model):
    """Stochastic SIR of vignettes/articles/stochastic-sir-model.Rmd:143-176,285-310: state (s, i), one Gillespie
    day per transition (rates lambda/n_total * s * i and gamma * i), y ~ Poisson(i).  `n_total` and the initial
    state are the vignette's globals (:143-148); the sampled parameters are (lambda, gamma)."""
    PARAM_ORDER = ("lambda", "gamma")

    def __init__(self, n_total=500, init_infected=70):
        self.name = "sir"
        self.init_fn = ModelFn("sir", "init", ())
        self.transition_fn = ModelFn("sir", "transition", ("lambda", "gamma"))
        self.log_likelihood_fn = ModelFn("sir", "log_likelihood", ())
        # look-ahead used by auxiliary_filter (the reference defines none for this model): Poisson at the one-day mean of i
        self.aux_log_likelihood_fn = ModelFn("sir", "aux_log_likelihood", ("lambda", "gamma"))
        self.dim = 2
        self.param_order = self.PARAM_ORDER
        self.constants = (float(n_total), float(n_total - init_infected), float(init_infected))
        for fn in (self.init_fn, self.transition_fn, self.log_likelihood_fn, self.aux_log_likelihood_fn):
            fn.owner = self


def sir(n_total=500, init_infected=70):
    return SirModel(n_total, init_infected)


class MoveFn:
    """Built-in move_fn of resample_move_filter: the random-walk Metropolis move of the reference's own example
    (R/resample_move_filter.R:166-176):  proposal = particle + rnorm(1, 0, sd);  accept when
    log(runif(1)) < log_likelihood(proposal) - log_likelihood(particle)."""

    def __init__(self, model, sd=0.1):
        self.model, self.sd = model, float(sd)

    def formals(self):
        return ["particle", "y", "sigma_y"]


class LinearGaussianMV:
    """Multivariate linear-Gaussian family on the device (state dimension d <= 8, observation dimension p <= 8):

        init_fn           x0 = m0 + L0 z               (the reference's  matrix(rnorm(N d), ncol = d)  shifted and scaled)
        transition_fn     x' = A x + b + L z           L lower triangular: a Cholesky factor of the state noise covariance
        log_likelihood_fn p == 0: the constant c0      (tests/testthat/test-bootstrap_filter.R:211-230: rep(1, nrow(particles)))
                          p >  0: sum_k dnorm(y_k, h0_k + (H x)_k, sd_k, log = TRUE)

    Fixed pieces are given to the constructor (m0, P0 or L0, A, b, Q or L, c0, H, h0, sd); pieces that depend on sampled
    parameters come from `build(**params) -> dict of pieces` (e.g. the reference's multi-dimensional PMMH case,
    tests/testthat/test-pmmh.R:619-668:  linear_gaussian_mv(2, build=lambda phi: {"b": [phi, phi]}, param_names=("phi",))).
    The three descriptors carry the parameter names, so bootstrap_filter / pmmh take them as they take the scalar models."""

    def __init__(self, d, p=0, build=None, param_names=(), **pieces):
        import numpy as np
        if not (1 <= int(d) <= 8 and 0 <= int(p) <= 8):
            raise ValueError("linear_gaussian_mv: 1 <= d <= 8 and 0 <= p <= 8")
        self.name, self.dim, self.p = "lgmv", int(d), int(p)
        self.build, self.param_order, self.constants = build, tuple(param_names), ()
        self.pieces = {"m0": np.zeros(self.dim), "L0": np.eye(self.dim), "A": np.eye(self.dim), "b": np.zeros(self.dim), "L": np.eye(self.dim),
                       "c0": 0.0, "H": np.eye(self.p, self.dim), "h0": np.zeros(self.p), "sd": np.ones(self.p)}
        self._set(pieces)
        self.init_fn = ModelFn("lgmv", "init", ())
        self.transition_fn = ModelFn("lgmv", "transition", self.param_order)
        self.log_likelihood_fn = ModelFn("lgmv", "log_likelihood", ())
        for fn in (self.init_fn, self.transition_fn, self.log_likelihood_fn):
            fn.owner = self

    def _set(self, pieces, into=None):
        import numpy as np
        tgt = self.pieces if into is None else into
        for k, v in pieces.items():
            if k == "P0":
                tgt["L0"] = np.linalg.cholesky(np.atleast_2d(np.asarray(v, dtype=np.float64)))
            elif k == "Q":
                tgt["L"] = np.linalg.cholesky(np.atleast_2d(np.asarray(v, dtype=np.float64)))
            elif k in self.pieces:
                tgt[k] = float(v) if k == "c0" else np.asarray(v, dtype=np.float64)
            else:
                raise TypeError("linear_gaussian_mv: unknown piece %r" % k)

    def pack(self, params):
        """the packed parameter block of include/bayesssm_amd.h (BSSM_MODEL_LGMV) for one parameter draw"""
        import numpy as np
        q = dict(self.pieces)
        if self.build is not None:
            missing = [k for k in self.param_order if k not in params]
            if missing:
                raise TypeError('argument "%s" is missing, with no default' % missing[0])
            self._set(self.build(**{k: float(params[k]) for k in self.param_order}), into=q)
        d, p = self.dim, self.p
        shapes = {"m0": (d,), "L0": (d, d), "A": (d, d), "b": (d,), "L": (d, d), "H": (p, d), "h0": (p,), "sd": (p,)}
        parts = [np.array([d, p], dtype=np.float64)]
        for k in ("m0", "L0", "A", "b", "L"):
            parts.append(np.broadcast_to(np.asarray(q[k], dtype=np.float64), shapes[k]).reshape(-1))
        parts[2] = np.tril(parts[2].reshape(d, d)).reshape(-1)          # (lower triangles: what a Cholesky factor is)
        parts[5] = np.tril(parts[5].reshape(d, d)).reshape(-1)
        parts.append(np.array([q["c0"]], dtype=np.float64))
        for k in ("H", "h0", "sd"):
            parts.append(np.broadcast_to(np.asarray(q[k], dtype=np.float64), shapes[k]).reshape(-1))
        return np.ascontiguousarray(np.concatenate(parts))


def linear_gaussian_mv(d, p=0, build=None, param_names=(), **pieces):
    return LinearGaussianMV(d, p, build, param_names, **pieces)


def linear_gaussian():
    """x0 ~ N(0,1); x' = phi x + N(0, sigma_x); y ~ N(x, sigma_y)
    (tests/testthat/test-pmmh_tuning.R:163-173 with free sigma_x, sigma_y; BASELINE C2/C3/C5)."""
    return Model("lg", linear_gaussian.__doc__)


def ar1_sin():
    """x0 ~ N(0,1); x' = phi x + sin(x) + N(0, sigma_x); y ~ N(x, sigma_y)   (README.md:137-146; BASELINE C1)."""
    return Model("ar1sin", ar1_sin.__doc__)


def resolve(init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn=None):
    """Check that the functions are descriptors of ONE built-in model and return its name."""
    fns = [("init_fn", init_fn, "init"), ("transition_fn", transition_fn, "transition"),
           ("log_likelihood_fn", log_likelihood_fn, "log_likelihood")]
    if aux_log_likelihood_fn is not None:
        fns.append(("aux_log_likelihood_fn", aux_log_likelihood_fn, "aux_log_likelihood"))
    for name, fn, role in fns:
        if not isinstance(fn, ModelFn):
            raise TypeError(
                "%s must be a built-in model descriptor from bayesssm_amd.models (arbitrary closures cannot run "
                "on the GPU; see DESIGN.md, 'user closures')" % name)
        if fn.role != role:
            raise ValueError("%s is a %s function, expected %s" % (name, fn.role, role))
    names = {fn.model for _, fn, _ in fns}
    if len(names) != 1:
        raise ValueError("init_fn, transition_fn and log_likelihood_fn belong to different models: %s" % sorted(names))
    return names.pop()


def theta_from_kwargs(fns, kwargs):
    """Collect the model's parameter vector from the named arguments the functions read
    (lg / ar1sin: phi, sigma_x, sigma_y;  sir: lambda, gamma, then the constants n_total, s0, i0)."""
    kwargs = dict(kwargs)
    if "lambda_" in kwargs:                       # `lambda` is a Python keyword
        kwargs["lambda"] = kwargs.pop("lambda_")
    needed = []
    for fn in fns:
        for p in fn.params:
            if p not in needed:
                needed.append(p)
    for p in needed:
        if p not in kwargs:
            raise TypeError('argument "%s" is missing, with no default' % p)   # R's message for a missing closure arg
    owner = getattr(fns[0], "owner", None)
    if owner is not None:
        return [float(kwargs.get(p, 1.0)) for p in owner.param_order] + list(owner.constants)
    return [float(kwargs.get(p, 1.0)) for p in Model.PARAM_ORDER]


def dim_of(model_name):
    return 2 if model_name == "sir" else 1            # ("lgmv": the dimension is the descriptor's, see filters.particle_filter_core)
