"""CPU oracle: test infrastructure only (see vgl_oracle.h). Never imported by the product package."""
