s1 s135 s5 s18
s17 s99 s73 s36 s0 s1 s38 s5 s9 s1 s114 s0 s48
s0 s7 s0 s6 s0 s2 s49 s1 s17 s5 s5
s21 s2 s48 s2 s0 s3 s26 OOVWORD s52 s11 s0 s46 s11 s22 s0 s72
s42 s65 s79 s0 s1 s7 s133 s18
s3 s64 s4 s149 s88 s22 s61 s2 s34 s0 s149 s51
s1 s42 s8 s0 s18 s153 s0 s40 s38 s60 s8 s1 s21 s0 s83 s52 s2 s27 s121 s0 s123 s39 s41 s5 s7 s91 s1 s9 s34 s3 s7 s0 s4 s3 s31 s63 s60 s1 s0 s24 s34 s46 s92 s3 s0 s4 s2 s0 s25 s9 s6 s4 s2 s5 s1 s71 s99 s6 s0 s2 s26 s142 s39 s0 s3 s0 s12 s61 s11 s19 s20 s6 s4 s141 s0 s75 s0 s0 s43 s0 s23 s1 s71 s13 s0 s149 s123 s0 s25 s26 s3 s5 s1 s9 s88 s7 s3 s7 s5 s5 s9 s32 s31 s0 s114 s3 s90 s104 s1 s102 s4 s25 s122 s15 s14 s0 s15 s1 s4 s4 s137 s20 s59 s5 s61 s0 s2 s153 s0 s44 s12 s9 s6 s1 s10 s9 s19 s33 s11 s0
