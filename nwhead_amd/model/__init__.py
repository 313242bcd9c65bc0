"""`load_model(name, pretrained=False, **kw)` with the reference's contract (model/__init__.py:5-8)."""


def load_model(name, pretrained=False, **kwargs):
    from . import backbones
    try:
        factory = getattr(backbones, name)
    except AttributeError:
        raise KeyError(name)
    return factory(pretrained=pretrained, **kwargs)
