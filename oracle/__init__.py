"""CPU checkers (test infrastructure only) -- see ssw_oracle.c."""
