from . import _utils, mobilenetv3  # noqa: F401
