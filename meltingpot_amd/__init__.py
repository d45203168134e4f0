"""meltingpot_amd: MI355X-native batched substrate engine for Melting Pot.

Replaces the DMLab2D/Lua step+render hot path behind
`meltingpot.substrate.build()` (reference: meltingpot/substrate.py:57,
utils/substrates/builder.py:142-192) with hand-written HIP kernels for gfx950.
"""

__version__ = "0.1.0"
