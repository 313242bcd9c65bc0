"""`load_model(name, pretrained=False, **kw)` with the reference's contract (model/__init__.py:5-8)."""


def load_model(name, pretrained=False, **kwargs):
    from . import backbones
    try:
        factory = getattr(backbones, name)
    except AttributeError:
        raise KeyError(name)
    return factory(pretrained=pretrained, **kwargs)


def fold_batchnorm(model):
    """Inference copy of a backbone with conv -> BatchNorm pairs folded (backbones.fold_batchnorm)."""
    from . import backbones
    return backbones.fold_batchnorm(model)
