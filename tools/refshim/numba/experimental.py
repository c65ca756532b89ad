def jitclass(spec=None):
    if isinstance(spec, type):
        return spec
    return lambda cls: cls
