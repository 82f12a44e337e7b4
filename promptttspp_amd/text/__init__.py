"""Text front-end tables (reference: promptttspp/text/)."""
