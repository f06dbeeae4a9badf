"""Parity and host-logic tests of the MI355X basecalling path (pytest; `-m gpu` needs a device)."""
