_entrypoints = {}


def register_model(fn):
    _entrypoints[fn.__name__] = fn
    return fn
