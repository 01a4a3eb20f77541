"""
dynamite_amd -- MI355X-native matrix-free spin-operator multiply and Krylov
engine, a drop-in for the ``dynamite.operators`` / ``dynamite.states`` API on
the ``Operator.dot`` / ``Operator.evolve`` / ``Operator.eigsolve`` path.
"""
from .config import config

__version__ = "0.1.0"
__all__ = ["config"]
