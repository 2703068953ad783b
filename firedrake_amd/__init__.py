"""firedrake_amd -- MI355X (gfx950) native finite-element assembly behind Firedrake's
assemble() / pyop2.parloop API.  See DESIGN.md."""
from . import op2  # noqa: F401

__version__ = "0.1.0"
