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
    """Collect (phi, sigma_x, sigma_y) from the named arguments the functions read."""
    needed = []
    for fn in fns:
        for p in fn.params:
            if p not in needed:
                needed.append(p)
    for p in needed:
        if p not in kwargs:
            raise TypeError('argument "%s" is missing, with no default' % p)   # R's message for a missing closure arg
    return [float(kwargs.get(p, 1.0)) for p in Model.PARAM_ORDER]
