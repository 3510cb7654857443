"""TEST-ONLY CPU oracle (see quadrace_oracle.c). Never imported by the product package."""
