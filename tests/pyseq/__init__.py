"""Python sequences of the one-call provers: the tests' second implementation (not part of the product package)."""
