"""Import alias: the product package directory is `alpha-zero-general_amd/` (not a valid Python identifier), so this
stub makes its modules importable as `azg_amd.<module>`."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'alpha-zero-general_amd')]
exec(open(os.path.join(__path__[0], '__init__.py')).read())
