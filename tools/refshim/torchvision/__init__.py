from . import models, ops  # noqa: F401
