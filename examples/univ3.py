"""Uniswap v3 pool -- the reference's examples/Univ3.jl on the MI355X path: one pool's arbitrage
at an external price of 25 (ν₁/ν₂ = 25)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cfmmrouter_amd as cr


def main():
    current_price = 15.0
    lower_ticks = [30.0, 20, 10, 5]
    liquidity = [1.0, 2.0, 1.5, 0.0]
    Ai = [1, 2]
    γ = 0.997
    cfmm = cr.UniV3(current_price, lower_ticks, liquidity, γ, Ai)

    Δ, Λ = np.zeros(2), np.zeros(2)
    cr.find_arb_(Δ, Λ, cfmm, [25.0, 1.0])
    eps = np.finfo(float).eps
    print("\tTendered basket:", ", ".join(f"{Ai[k]}: {round(δ, 3)}" for k, δ in enumerate(Δ) if δ > eps))
    print("\tReceived basket:", ", ".join(f"{Ai[k]}: {round(λ, 3)}" for k, λ in enumerate(Λ) if λ > eps))
    return Δ, Λ


if __name__ == "__main__":
    main()
