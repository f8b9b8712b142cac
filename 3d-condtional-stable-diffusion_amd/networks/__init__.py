"""Mirrors the reference's ``networks`` package for the two modules on the denoising path."""
