"""euclider_amd -- MI355X-native implementation of euclider's per-pixel trace loop.

Only the hot path (Environment::render -> Universe::trace -> Entity::intersect recursion) lives
here, behind the C ABI of include/euclider_amd.h; see DESIGN.md.
"""
from .environment import Environment, EuError, FrameSequence, Parser, ParserError, RawImage2d, SimulationContext  # noqa: F401

__all__ = ["Environment", "EuError", "FrameSequence", "Parser", "ParserError", "RawImage2d", "SimulationContext"]
